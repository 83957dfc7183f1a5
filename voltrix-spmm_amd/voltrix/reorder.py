"""Locality reorder of a graph (the "reorder" half).

Two layers.  (1) Host utilities from round 1 (scipy RCM, symmetric permutation ``P A P^T``; the caller permutes B and
un-permutes C).  (2) Round 2, the product form: ``csr_preprocess_reordered`` / ``spmm_reordered`` -- everything on the GPU
and carried by the handle.  Only the ROWS of A are regrouped (which 16 rows share a window decides the TC-block fill;
column ids are not relabelled), so B is gathered as the caller has it, and the SpMM writes row i of the handle to row
``row_map[i]`` of C through the kernel's epilogue: no permute pass before, no un-permute pass after.  The row order comes
from a level-synchronous breadth-first search of the pattern of A + A^T on the device (Cuthill-McKee levels from a
pseudo-peripheral start; round 3: HIP kernels, include/voltrix/reorder_kernels.hpp), from a spectral embedding computed with
the SpMM kernels, from the degrees, or from the caller.

--- round 1 docstring ---
Locality reorder of a graph before ``csr_preprocess``.

The reference has no reordering code: its benchmark reads externally reordered ``<name>.reorder.npz`` files
(bench/graph_gen.py:42-45, bench/bench_all.py:120-129).  A symmetric permutation ``P A P^T`` that pulls the non-zeros
towards the diagonal raises the fill of the 16 x 8 TC blocks (fewer gathered rows of B per window) and makes the
windows that run together share their columns through L2.  This module offers the classic reverse Cuthill-McKee order
(scipy, host side -- like the reference's own host preprocess it is a one-time cost) and the bookkeeping around it:

    perm = reorder.rcm_permutation(indptr, indices, n)          # new position k holds old node perm[k]
    indptr2, indices2 = reorder.permute_csr(indptr, indices, n, perm)
    handle = voltrix.csr_preprocess(indptr2, indices2, n)
    out = reorder.unpermute_rows(voltrix.spmm(*handle, n, e, reorder.permute_rows(feat, perm)), perm)

``out`` equals ``voltrix.spmm`` on the original graph up to fp32 summation order.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse.csgraph import reverse_cuthill_mckee


def _csr(indptr, indices, num_nodes):
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    return sp.csr_matrix((np.ones(indices.shape[0], dtype=np.int8), indices, indptr), shape=(num_nodes, num_nodes))


def rcm_permutation(indptr, indices, num_nodes: int) -> np.ndarray:
    """Reverse Cuthill-McKee order of the symmetrised pattern; ``perm[k]`` = old id of the node placed at position k."""
    a = _csr(indptr, indices, num_nodes)
    return np.asarray(reverse_cuthill_mckee((a + a.T).tocsr(), symmetric_mode=True), dtype=np.int64)


def degree_permutation(indptr, num_nodes: int) -> np.ndarray:
    """Nodes by descending out-degree (a cheap alternative when the graph has no recoverable band structure)."""
    deg = np.diff(np.asarray(indptr, dtype=np.int64))[:num_nodes]
    return np.argsort(-deg, kind="stable").astype(np.int64)


def permute_csr(indptr, indices, num_nodes: int, perm: np.ndarray):
    """CSR of ``P A P^T`` (rows and columns relabelled), int32, rows sorted, duplicates kept."""
    a = _csr(indptr, indices, num_nodes)
    b = a[perm][:, perm].tocsr()
    b.sort_indices()
    return torch.from_numpy(b.indptr.astype(np.int32)), torch.from_numpy(b.indices.astype(np.int32))


def permute_rows(x: torch.Tensor, perm: np.ndarray) -> torch.Tensor:
    """Rows of ``x`` in the new node order (``out[k] = x[perm[k]]``)."""
    return x[torch.as_tensor(perm, device=x.device)]


def unpermute_rows(x: torch.Tensor, perm: np.ndarray) -> torch.Tensor:
    """Inverse of :func:`permute_rows` (``out[perm[k]] = x[k]``)."""
    out = torch.empty_like(x)
    out[torch.as_tensor(perm, device=x.device)] = x
    return out


def bandwidth(indptr, indices, num_nodes: int) -> int:
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(num_nodes, dtype=np.int64), np.diff(indptr)[:num_nodes])
    return int(np.abs(rows - np.asarray(indices, dtype=np.int64)).max()) if rows.size else 0


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: reorder on the GPU, permutation carried by the handle (no reference counterpart: the reference reads
# externally reordered <name>.reorder.npz files, bench/graph_gen.py:42-45).
import dataclasses  # noqa: E402


def _cm_degrees(indptr: torch.Tensor, indices: torch.Tensor, n: int, num_cols: int):
    """deg(u) = entries of row u of A + entries of row u of A^T, and tie(u) = position of u in the stable sort by deg
    (the specification in voltrix/include/voltrix/reorder_kernels.hpp)."""
    deg = (indptr[1:] - indptr[:-1]).long()
    if indices.numel():
        into = torch.bincount(indices.long(), minlength=max(n, num_cols))[:n]
        into = torch.where(torch.arange(n, device=indptr.device) < num_cols, into, torch.zeros_like(into))
        deg = deg + into
    tie = torch.empty(n, dtype=torch.int64, device=indptr.device)
    tie[torch.argsort(deg, stable=True)] = torch.arange(n, device=indptr.device)
    return deg, tie


def _both_directions(indptr: torch.Tensor, indices: torch.Tensor, n: int):
    """Host form of the search's neighbour lists: CSR over the nodes whose row u lists row u of A (columns < n) followed
    by row u of A^T -- nothing de-duplicated, like the kernels."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
    cols = indices.long()
    keep = cols < n
    rows, cols = rows[keep], cols[keep]
    su, sv = torch.cat([rows, cols]), torch.cat([cols, rows])
    order = torch.argsort(su, stable=True)
    sptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    sptr[1:] = torch.cumsum(torch.bincount(su, minlength=n), 0)
    return sptr, sv[order]


def _bfs_levels(sptr, sv, tie, start: int, level, rank, next_rank: int):
    """Level-synchronous search from ``start`` over the unvisited part; fills ``level`` and ``rank`` (inside a level by the
    rank of the earliest-ranked neighbour of the level before, then by ``tie``).  Returns (next_rank, last level)."""
    dev = sptr.device
    n = tie.numel()
    sdeg = sptr[1:] - sptr[:-1]
    frontier = torch.tensor([start], device=dev, dtype=torch.int64)
    level[start] = 0
    rank[start] = next_rank
    next_rank += 1
    d = 0
    last = frontier
    while frontier.numel():
        counts = sdeg[frontier]
        total = int(counts.sum())
        if total == 0:
            break
        base = torch.repeat_interleave(sptr[frontier] - (torch.cumsum(counts, 0) - counts), counts)
        nb = sv[base + torch.arange(total, device=dev)]
        parent_rank = torch.repeat_interleave(rank[frontier], counts)
        fresh = level[nb] < 0
        nb, parent_rank = nb[fresh], parent_rank[fresh]
        if nb.numel() == 0:
            break
        uniq, inv = torch.unique(nb, return_inverse=True)
        best_parent = torch.full((uniq.numel(),), 1 << 62, dtype=torch.int64, device=dev)
        best_parent.scatter_reduce_(0, inv, parent_rank, reduce="amin")
        uniq = uniq[torch.argsort(best_parent * n + tie[uniq])]
        d += 1
        level[uniq] = d
        rank[uniq] = next_rank + torch.arange(uniq.numel(), device=dev)
        next_rank += int(uniq.numel())
        last = frontier = uniq
    return next_rank, last


def _cm_permutation_host(indptr, indices, n, num_cols, max_components):
    """The search with torch tensor ops (CPU tensors: a host utility like ``rcm_permutation``; the device path below runs
    the HIP kernels)."""
    dev = indptr.device
    deg, tie = _cm_degrees(indptr, indices, n, num_cols)
    sptr, sv = _both_directions(indptr, indices, n)
    level = torch.full((n,), -1, dtype=torch.int64, device=dev)
    rank = torch.full((n,), -1, dtype=torch.int64, device=dev)
    next_rank = 0
    for _ in range(max_components):
        cand = torch.where((level < 0) & (deg > 0), tie, torch.full_like(tie, n))
        start = int(torch.argmin(cand))
        if int(cand[start]) >= n:
            break
        # pseudo-peripheral start: search once, restart from the node of the last level that comes first in the tie order
        probe_level, probe_rank = level.clone(), rank.clone()
        _, last = _bfs_levels(sptr, sv, tie, start, probe_level, probe_rank, next_rank)
        start = int(last[torch.argmin(tie[last])])
        next_rank, _ = _bfs_levels(sptr, sv, tie, start, level, rank, next_rank)
    return deg, rank, next_rank


def _cm_permutation_device(indptr, indices, n, num_cols, max_components, info=None):
    """The search on the device: voltrix/include/voltrix/reorder_kernels.hpp through the C-ABI.  Host reads: one per
    component for the start node, one per search for its size, one for the level offsets -- not one per level."""
    from . import capi

    dev = indptr.device
    t_indptr, t_indices = capi.csr_transpose(indptr, indices, n, num_cols)
    deg = (indptr[1:] - indptr[:-1]).long()                  # _cm_degrees, with A^T's row pointers in place of a bincount
    k = min(n, num_cols)
    deg[:k] += (t_indptr[1:k + 1] - t_indptr[:k]).long()
    tie = torch.empty(n, dtype=torch.int64, device=dev)
    tie[torch.argsort(deg, stable=True)] = torch.arange(n, device=dev)
    search = capi.CmSearch(indptr, indices, t_indptr, t_indices, n, num_cols, tie.to(torch.int32))
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    base = 0
    components = 0
    for _ in range(max_components):
        cand = torch.where((search.level < 0) & (deg > 0), tie, torch.full_like(tie, n))
        start = int(torch.argmin(cand))
        if int(cand[start]) >= n:
            break
        nodes, levels = search.levels(start)
        last = search.queue[int(search.level_off[levels - 1]):nodes].long()
        start = int(last[torch.argmin(tie[last])])
        search.level[search.queue[:nodes].long()] = -1
        nodes, levels = search.levels(start)
        search.rank_component(levels, base)
        perm[base:base + nodes] = search.queue[:nodes]
        base += nodes
        components += 1
    if info is not None:
        info.update(components=components, level_reads=search.syncs)
    return deg, search.rank.long(), base, perm


def bfs_permutation(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                    max_components: int = 64, info: dict = None) -> torch.Tensor:
    """Cuthill-McKee row order: int64 [N], position k holds row ``perm[k]``.  Breadth-first levels of the pattern of
    A + A^T from a pseudo-peripheral start (two searches), component by component (``max_components`` of them, each started
    from the unvisited node of least degree; what is left -- isolated rows, further components -- follows by descending
    degree).  Rows that are close in the graph end up in the same 16-row windows and share gathered columns.  Device CSR: HIP
    kernels (reorder_kernels.hpp -- frontier expansion with atomics, per-level key sort; the specification is in that
    header); CPU CSR: the same specification with torch tensor ops.  Both give the same permutation."""
    n = num_nodes
    dev = indptr.device
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    num_cols = n if num_cols is None else int(num_cols)
    if indptr.is_cuda:
        deg, rank, next_rank, perm = _cm_permutation_device(indptr.contiguous(), indices.contiguous(), n, num_cols,
                                                            max_components, info)
    else:
        deg, rank, next_rank = _cm_permutation_host(indptr, indices, n, num_cols, max_components)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
    rest = torch.nonzero(rank < 0).flatten()
    if rest.numel():
        rest = rest[torch.argsort(-deg[rest], stable=True)]
        rank[rest] = next_rank + torch.arange(rest.numel(), device=dev)
    if indptr.is_cuda:
        perm[next_rank:] = rest
    else:
        perm[rank] = torch.arange(n, device=dev)
    return perm


# Unfolding of the spectral order (round 5).  Cells of the recursive bisection of the leading eigenvectors' embedding, and how many of
# those vectors are used.  Graphs below UNFOLD_MIN_ROWS keep the plain sort by the leading vector.
UNFOLD_CELLS = 2048
UNFOLD_VECTORS = 8
UNFOLD_MIN_ROWS = 1 << 16
UNFOLD_ALL_PAIRS_CELLS = 4096  # up to this many boxes: hop distances between all of them; above: to 256 landmark boxes
UNFOLD_ROWS_PER_CELL = 160     # graphs of more than 2048 x this many rows get more boxes (up to 16384) and 16 leading vectors
UNFOLD_MAX_ROWS = 1 << 22      # the neighbour votes hold a [rows, 512] fp32 matrix (8.6 GB here); larger graphs keep the plain sort
UNFOLD_MIN_ONE_DIMENSIONAL = 30.0   # leading / second eigenvalue of the boxes' distance scaling below which the unfolding is dropped
SUBSPACE_SPREAD = 0.8          # more subspace-iteration rounds while the smallest Ritz value is above this x the largest ...
SUBSPACE_EXTRA_ROUNDS = 5      # ... at most this many rounds of `iterations` steps


def unfolded_order(coords: torch.Tensor, indptr: torch.Tensor, indices: torch.Tensor, active: torch.Tensor,
                   cells: int = UNFOLD_CELLS, return_info: bool = False):
    """Row order of a SQUARE graph from a k-dimensional spectral embedding ``coords`` [N, k] (round 5): int64 [N].

    On a band graph with a background of random edges the leading eigenvectors of the co-occurrence operator are nearly
    degenerate (reddit-like stand-in: 0.2360, 0.2349, 0.2335, 0.2320 -- the gap goes with (band / N)^2), and the one that comes out
    on top is a MIXTURE of the first harmonics: not monotone along the band.  Sorting by it interleaves two or three distant
    stretches of the band in every window (measured against the known generating order: correlation 0.67,
    harness/experiments/exp_spectral_error.py).  Any single eigenvector of a LOCAL operator has that problem, at every scale.
    The SUBSPACE of the leading vectors is stable though: in k dimensions the band is a curve that does not cross itself, and
    DISTANCE ALONG THE CURVE is a global quantity with no near-degeneracy (the idea of Isomap):
      1. cut the embedding into ``cells`` boxes by recursive median bisection along each box's widest coordinate; a box
         normally holds one short stretch of the curve; boxes that caught two stretches are recognised by their extent
         (several times the median) and set aside;
      2. count the edges between boxes; with ~100 rows per box the random background averages out (its expectation, the
         rank-one term d d^T / sum d, is subtracted); every box keeps its strongest links (a band: its neighbours on the curve);
      3. hop distances between all boxes over those links (dense boolean products), classical multidimensional scaling of the
         distance matrix to ONE coordinate (leading eigenvector of the double-centred squared distances, float64): the order
         of the boxes along the curve;
      4. rows sorted by the rank of their box (stable); rows of the boxes set aside, and every row's place inside the order,
         are settled by the caller's refinement (neighbour votes).
    Rows outside ``active`` (no edges) go last."""
    import time as _time

    dev = coords.device
    n, k = coords.shape
    levels = max(1, int(cells).bit_length() - 1)
    marks = []

    def mark(name):
        if return_info and coords.is_cuda:
            torch.cuda.synchronize()
        marks.append((name, _time.perf_counter()))

    mark("start")
    act_idx = torch.nonzero(active).flatten()
    na = act_idx.numel()
    y = coords[act_idx].double()
    y = (y - y.mean(0)) / y.std(0).clamp(min=1e-30)
    # rows kept SORTED BY BOX (order): a box is a contiguous segment, its sums are differences of running sums (no atomics:
    # at the first levels millions of rows would add into a handful of addresses)
    order = torch.arange(na, device=dev)
    bounds = torch.tensor([0, na], dtype=torch.int64, device=dev)       # segment b = order[bounds[b] : bounds[b + 1]]

    yt = y.T.contiguous()                                                # [k, rows]: scans run along the contiguous dimension

    def segment_stats(ys, bounds):                                       # ys [k, rows in box order]
        size = (bounds[1:] - bounds[:-1]).clamp(min=1).double()[None, :]
        cs = torch.nn.functional.pad(ys.cumsum(1), (1, 0))
        cs2 = torch.nn.functional.pad((ys * ys).cumsum(1), (1, 0))
        mean = (cs[:, bounds[1:]] - cs[:, bounds[:-1]]) / size
        var = ((cs2[:, bounds[1:]] - cs2[:, bounds[:-1]]) / size - mean * mean).clamp(min=0.0)
        return mean.T, var.T

    for _ in range(levels):
        g = bounds.numel() - 1
        ys = yt[:, order]
        _, var = segment_stats(ys, bounds)
        seg = torch.repeat_interleave(torch.arange(g, device=dev), bounds[1:] - bounds[:-1])
        dim = var.argmax(1)                                  # every box splits along its widest coordinate ...
        v = ys.gather(0, dim[seg][None, :])[0]
        lo, hi = v.min(), v.max()
        q = ((v - lo) / ((hi - lo) + 1e-300) * float((1 << 31) - 1)).long()      # 31-bit quantised coordinate
        order = order[torch.argsort((seg << 32) | q)]        # ... at its median: box, then the coordinate inside it
        mid = (bounds[:-1] + bounds[1:] + 1) // 2
        bounds = torch.stack([bounds[:-1], mid], dim=1).flatten()
        bounds = torch.cat([bounds, torch.tensor([na], dtype=torch.int64, device=dev)])
    mark("bisection")
    g = bounds.numel() - 1
    size = bounds[1:] - bounds[:-1]
    ys = yt[:, order]
    _, var = segment_stats(ys, bounds)
    radius = var.sum(1).sqrt()
    pure = (size > 0) & (radius <= 3.0 * radius[size > 0].median())
    c = torch.empty(na, dtype=torch.int64, device=dev)
    c[order] = torch.repeat_interleave(torch.arange(g, device=dev), size)
    cell = torch.full((n,), -1, dtype=torch.int64, device=dev)
    cell[act_idx] = c
    # edges between boxes (square graph: a column id is a row)
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
    keep = active[rows] & active[indices.long()]
    pair = (cell[rows] * g + cell[indices.long()])[keep]
    del rows, keep
    w = torch.bincount(pair, minlength=g * g).view(g, g).double()
    del pair
    mark("edges between boxes")
    w = w + w.T
    d = w.sum(1)
    background = torch.outer(d, d) / d.sum().clamp(min=1.0)  # the uniform background's expectation
    w = w - background
    w.fill_diagonal_(0.0)
    w[~pure, :] = 0.0
    w[:, ~pure] = 0.0
    # a box's links: its strongest ones (at most 16), as long as they stand clear of the background's noise (five standard
    # deviations of its count) and are not dwarfed by the box's best link -- a band narrower than 16 boxes must not pick up
    # chance links, one shortcut would shorten every distance across it -- and only when BOTH boxes name each other (with
    # either-names-the-other, or with a further purity test by the neighbours' mutual links, both stand-ins came out worse:
    # reddit-like correlation 0.39, products-like hop count 9).
    links = min(16, g - 1)
    top = torch.topk(w, links, dim=1)
    strong = (top.values >= 5.0 * (background.gather(1, top.indices) + 1.0).sqrt()) & (top.values >= 0.15 * top.values[:, :1])
    adj = torch.zeros(g, g, dtype=torch.bool, device=dev)
    adj[torch.arange(g, device=dev)[:, None].expand(-1, links)[strong], top.indices[strong]] = True
    adj = adj & adj.T                                        # both boxes name each other
    # hop distances over the links
    hops = 0
    if g <= UNFOLD_ALL_PAIRS_CELLS:
        # all boxes at once: frontier expansion by dense boolean products
        reach = torch.eye(g, dtype=torch.bool, device=dev)
        dist = torch.full((g, g), float("inf"), dtype=torch.float32, device=dev)
        dist[reach] = 0.0
        adj_f = adj.float()
        while hops < g:
            hops += 1
            nxt = ((reach.float() @ adj_f) > 0) & ~reach
            if not bool(nxt.any()):
                break
            dist[nxt] = float(hops)
            reach |= nxt
        mark("hop distances")
        # the largest connected set of pure boxes carries the order; the others are placed by the neighbour votes
        comp = reach[int(reach.sum(1).argmax())] & pure
        idx = torch.nonzero(comp).flatten()
        dd = dist[idx][:, idx].double() ** 2
        j = dd - dd.mean(0, keepdim=True) - dd.mean(1, keepdim=True) + dd.mean()
        evals, evecs = torch.linalg.eigh(-0.5 * j)
        coord = evecs[:, -1]
    else:
        # many boxes (long thin graphs): distances to LANDMARK boxes only, and landmark scaling (de Silva & Tenenbaum): the
        # leading eigenpair of the landmarks' own double-centred squared distances, every other box by triangulation
        num_landmarks = 256
        gen = torch.Generator(device=dev)
        gen.manual_seed(0)
        cand = torch.nonzero(pure & (adj.sum(1) > 0)).flatten()
        lm = cand[torch.randperm(cand.numel(), generator=gen, device=dev)[:num_landmarks]]
        num_landmarks = lm.numel()
        reach = torch.zeros(g, num_landmarks, dtype=torch.bool, device=dev)
        reach[lm, torch.arange(num_landmarks, device=dev)] = True
        dist = torch.full((g, num_landmarks), float("inf"), dtype=torch.float32, device=dev)
        dist[reach] = 0.0
        adj_h = adj.half()
        while hops < g:
            hops += 1
            nxt = ((adj_h @ reach.half()) > 0) & ~reach      # at most 16 links per box: the sums are exact in fp16
            if not bool(nxt.any()):
                break
            dist[nxt] = float(hops)
            reach |= nxt
        mark("hop distances")
        # landmarks of the largest connected set, and the boxes that set reaches
        ll = reach[lm]                                        # [L, L]: landmark i reached from landmark j
        main = ll[:, int(ll.sum(0).argmax())]
        keep = torch.nonzero(main).flatten()
        comp = reach[:, keep].all(1) & pure
        idx = torch.nonzero(comp).flatten()
        dl = dist[lm[keep]][:, keep].double() ** 2            # landmark-landmark squared distances
        mean_col = dl.mean(0)
        j = dl - dl.mean(0, keepdim=True) - dl.mean(1, keepdim=True) + dl.mean()
        evals, evecs = torch.linalg.eigh(-0.5 * j)
        v1, l1 = evecs[:, -1], evals[-1].clamp(min=1e-30)
        dx = dist[idx][:, keep].double() ** 2
        coord = -0.5 * ((dx - mean_col[None, :]) @ v1) / l1.sqrt()
    mark("scaling")
    cell_pos = torch.full((g,), float("nan"), dtype=torch.float64, device=dev)
    cell_pos[idx] = torch.argsort(torch.argsort(coord)).double()
    # boxes outside the ordered set: next to their strongest ordered link (or last)
    rest = torch.nonzero(~comp & (size > 0)).flatten()
    cell_pos[rest] = float(idx.numel())                      # after the ordered boxes; neighbour_votes places their rows
    key = torch.where(cell >= 0, cell_pos[cell.clamp(min=0)], torch.full((n,), float("inf"), dtype=torch.float64, device=dev))
    perm = torch.argsort(key, stable=True)
    if return_info:
        return perm, {"ms": {b_[0]: round((b_[1] - a_[1]) * 1e3, 1) for a_, b_ in zip(marks, marks[1:])}, "cells": g,
                      "one_dimensional": float(evals[-1] / evals[-2].abs().clamp(min=1e-30)) if idx.numel() > 1 else 0.0, "pure": int(pure.sum()), "ordered": int(idx.numel()), "hops": hops,
                      "mds_top_eigenvalues": evals[-3:].flip(0).tolist(), "settled": comp[cell.clamp(min=0)] & (cell >= 0)}
    return perm


def neighbour_votes(apply_a, perm: torch.Tensor, settled: torch.Tensor, active: torch.Tensor, buckets: int = 256,
                    reach: int = 3, degree: torch.Tensor = None, confidence: float = 0.1) -> torch.Tensor:
    """One round of neighbour votes (round 5): every row moves to where most of its neighbours are.  ``perm`` orders the
    ``settled`` rows (they come first); their positions are cut into ``buckets`` equal stretches, ``apply_a(B)`` = ``A @ B`` of the
    one-hot bucket matrix counts every row's neighbours per stretch (one SpMM, ``buckets`` columns), and the row goes to the
    centroid of the densest window of ``2 reach + 1`` stretches: the local neighbours of a band graph stand together there while
    the random ones spread over all stretches.  Rows that were not settled (boxes the unfolding set aside) are placed the same
    way; rows without a settled neighbour keep their place.  Returns the new order (inactive rows last)."""
    dev = perm.device
    n = perm.numel()
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[perm] = torch.arange(n, device=dev)
    ns = max(1, int(settled.sum()))
    b = (pos * buckets // ns).clamp(max=buckets - 1)
    onehot = torch.zeros(n, buckets, dtype=torch.float16, device=dev)
    rows = torch.nonzero(settled).flatten()
    onehot[rows, b[rows]] = 1.0
    votes = apply_a(onehot).float()                                     # [N, buckets] neighbour counts
    del onehot
    width = 2 * reach + 1
    cs = torch.nn.functional.pad(votes.cumsum(1), (1, 0))
    window = cs[:, width:] - cs[:, :-width]                             # window j = stretches j .. j + width - 1
    best = window.argmax(1)
    total = window.gather(1, best[:, None])[:, 0]
    offs = torch.arange(width, device=dev)
    idx = best[:, None] + offs[None, :]
    wts = votes.gather(1, idx)
    centre = (wts * (idx.float() + 0.5)).sum(1) / wts.sum(1).clamp(min=1e-30)
    sure = total > 0
    if degree is not None:   # a window that holds only chance votes moves nobody (argmax of noise: the first stretch)
        sure = (total >= 2.0) & (total >= confidence * degree)
    new_pos = torch.where(sure, centre * (ns / buckets), pos.float())
    key = torch.where(active, new_pos, torch.full_like(new_pos, float("inf")))
    return torch.argsort(key, stable=True)


def grow_settled(apply_a, perm: torch.Tensor, settled: torch.Tensor, active: torch.Tensor, degree: torch.Tensor,
                 buckets: int = 256, reach: int = 3, max_rounds: int = 256, confidence: float = 0.15):
    """Rows the unfolding could not order (boxes off the main chain: whole stretches of a long thin band) join it from its ends
    inwards (round 5): per round one SpMM counts every row's SETTLED neighbours per stretch of the settled order; an unsettled
    row whose densest window holds at least ``confidence`` of its edges (the uniform background puts ``(2 reach + 1) / buckets``
    of them into any window) is placed at that window's centroid and settles; its own neighbours find it there in the next
    round.  Settled rows keep their relative order.  Returns ``(perm, settled)``: settled rows first, in order."""
    dev = perm.device
    n = perm.numel()
    settled = settled.clone()
    pos = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
    ns = int(settled.sum())
    pos[perm[:ns]] = torch.arange(ns, device=dev, dtype=torch.float64)
    width = 2 * reach + 1
    offs = torch.arange(width, device=dev)
    for _ in range(max_rounds):
        todo = active & ~settled
        if not bool(todo.any()) or ns == 0:
            break
        rank = torch.empty(n, dtype=torch.int64, device=dev)
        order = torch.argsort(pos, stable=True)
        rank[order] = torch.arange(n, device=dev)
        b = (rank * buckets // ns).clamp(max=buckets - 1)
        onehot = torch.zeros(n, buckets, dtype=torch.float16, device=dev)
        rows = torch.nonzero(settled).flatten()
        onehot[rows, b[rows]] = 1.0
        votes = apply_a(onehot).float()
        del onehot
        cs = torch.nn.functional.pad(votes.cumsum(1), (1, 0))
        window = cs[:, width:] - cs[:, :-width]
        best = window.argmax(1)
        total = window.gather(1, best[:, None])[:, 0]
        idx = best[:, None] + offs[None, :]
        wts = votes.gather(1, idx)
        centre = (wts * (idx.float() + 0.5)).sum(1) / wts.sum(1).clamp(min=1e-30)
        sure = todo & (total >= 2.0) & (total >= confidence * degree)   # two votes in one window: 3 % by chance at 7 of 256 stretches
        if not bool(sure.any()):
            break
        # a stretch's centre in the positions of the settled order: (centre / buckets) ns, between the ranks of its neighbours
        pos[sure] = centre[sure].double() * (ns / buckets) - 0.5
        settled |= sure
        ns = int(settled.sum())
        order = torch.argsort(pos, stable=True)
        pos[order[:ns]] = torch.arange(ns, device=dev, dtype=torch.float64)    # ranks again: the next round's stretches
    key = torch.where(settled, pos, torch.full_like(pos, float("inf")))
    return torch.argsort(key, stable=True), settled


def spectral_permutation(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                         vectors: int = 32, iterations: int = 16, seed: int = 0, return_info: bool = False,
                         refine: int = 4, refine_width: int = 8192, unfold: bool = None, vote_rounds: int = 2):
    """Spectral row order on the device, computed WITH the SpMM kernels (round 3): int64 [N], position k holds row ``perm[k]``.

    Rows that reference the same columns should share a 16-row window (and a 512-row panel).  Their similarity is the
    co-occurrence matrix ``A A^T``; the leading non-trivial eigenvector of its normalised form
    ``K = D_r^-1/2 A D_c^-1 A^T D_r^-1/2`` (the Fiedler vector of the row side of the bipartite row-column graph) is a
    one-dimensional embedding in which such rows are neighbours, and sorting by it is the row order.  Unlike a
    breadth-first search it sees through a background of random edges: on the reddit-like stand-in half of the edges are
    uniformly random, every BFS level after the second holds the whole graph, while ``K`` only loses half of its spectrum's
    scale to them (a uniform background is a rank-one term along the trivial eigenvector).

    Block subspace iteration: ``vectors`` columns, each step two products -- ``A^T Z`` and ``A Y``, i.e. ``voltrix.spmm`` on
    the handle of ``A`` and on the handle of ``A^T`` (32 columns: the narrow-feature tiles) -- a deflation of the trivial
    eigenvector and a Cholesky QR; the error of the leading vector shrinks like ``(lambda_33 / lambda_2)^steps``, so two dozen
    steps are plenty where single-vector power iteration would need thousands.  Rayleigh-Ritz in the final subspace.  Column
    ids are not relabelled.  Rows without edges go last.  Deterministic for a given ``seed``."""
    from . import capi, hybrid
    from .autograd import csr_transpose_device
    from .jit_kernels import spmm as spmm_wrapper
    from .spmm.spmm import csr_preprocess_device, spmm

    import time as _time

    n = num_nodes
    dev = indptr.device
    assert indptr.is_cuda, "the spectral order runs the SpMM kernels: device CSR"
    stamps = []

    def stamp(name):
        if return_info:
            torch.cuda.synchronize()
            stamps.append((name, _time.perf_counter()))

    stamp("start")
    if n == 0:
        return (torch.zeros(0, dtype=torch.int64, device=dev), {}) if return_info else torch.zeros(0, dtype=torch.int64, device=dev)
    m = n if num_cols is None else int(num_cols)
    nnz = int(indices.numel())
    # the products below are plumbing of a one-time preprocess: default tiles, no tuning sweep, no side-car -- as overrides of
    # THIS context (other threads of the process keep what the environment says; rounds 2-3 rewrote os.environ here)
    unfold_info = None
    with spmm_wrapper.tune_space("none"), hybrid.mode_override("0"):
        handle = csr_preprocess_device(indptr, indices, n, num_cols=m)
        t_indptr, t_indices = csr_transpose_device(indptr, indices, n, m)
        handle_t = csr_preprocess_device(t_indptr, t_indices, m, num_cols=n)
        handle[1].hash_tag = handle_t[1].hash_tag = None   # untagged: keyed by address, never persisted
        stamp("handles of A and A^T")
        deg_r = (indptr[1:] - indptr[:-1]).float()
        deg_c = (t_indptr[1:] - t_indptr[:-1]).float()
        del t_indptr, t_indices
        ir = torch.where(deg_r > 0, deg_r.clamp(min=1).rsqrt(), torch.zeros_like(deg_r))[:, None]   # D_r^-1/2
        ic = torch.where(deg_c > 0, 1.0 / deg_c.clamp(min=1), torch.zeros_like(deg_c))[:, None]     # D_c^-1
        trivial = deg_r.sqrt()[:, None]
        trivial = trivial / trivial.norm()
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        x = torch.randn(n, vectors, generator=gen, device=dev)

        def apply_k(x):
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")      # the untagged-handle warning of feature_hash, once per call
                y = spmm(*handle_t, num_nodes=m, num_edges=nnz, feat=(x * ir).contiguous())
                return spmm(*handle, num_nodes=n, num_edges=nnz, feat=(y * ic).contiguous()) * ir

        def orthonormalise(x):
            x = x - trivial @ (trivial.T @ x)
            for _ in range(2):                     # Cholesky QR, twice: n x 32 against a 32 x 32 factor
                # the small factor on the DEVICE (reorder_kernels.hpp::chol_inv_transposed_kernel, one workgroup, double
                # arithmetic): no host sync per step (rounds 2-3: numpy on the host, two syncs per step)
                x = x @ capi.chol_inv_transposed((x.T @ x).contiguous(), 1e-10)
            return x

        x = orthonormalise(x)
        x = orthonormalise(apply_k(x))
        stamp("first product (kernel load, unit tables)")
        for _ in range(iterations - 1):
            x = orthonormalise(apply_k(x))
        kx = apply_k(x)
        proj = (x.T @ kx).double().cpu().numpy()             # Rayleigh-Ritz on the 32 x 32 projection (host: tiny), ascending
        evals_np, evecs_np = np.linalg.eigh(0.5 * (proj + proj.T))
        # a long thin band (N / band in the hundreds: products-like) keeps dozens of harmonics within a few per cent of the
        # leading eigenvalue, and the subspace has not separated them from the rest after the usual number of steps (the
        # error shrinks like (lambda_33 / lambda_2)^steps): keep going, in rounds, until the subspace's own spectrum has a
        # spread -- bounded (round 5; the unfolding below needs the leading vectors' SUBSPACE, not just one vector)
        extra_rounds = 0
        while (unfold is not False and evals_np[-1] > 0 and evals_np[0] > SUBSPACE_SPREAD * evals_np[-1]
               and extra_rounds < SUBSPACE_EXTRA_ROUNDS):
            extra_rounds += 1
            if extra_rounds == 1 and x.shape[1] < 2 * vectors:   # twice the vectors: the tail they must beat lies further down
                x = orthonormalise(torch.cat([x, torch.randn(n, vectors, generator=gen, device=dev)], dim=1))
            for _ in range(iterations):
                x = orthonormalise(apply_k(x))
            kx = apply_k(x)
            proj = (x.T @ kx).double().cpu().numpy()
            evals_np, evecs_np = np.linalg.eigh(0.5 * (proj + proj.T))
        stamp("subspace iteration")
        evals = torch.from_numpy(evals_np)
        fiedler = (x @ torch.from_numpy(evecs_np[:, -1:].astype(np.float32)).to(dev)) * ir   # random-walk coordinates
        key = torch.where(deg_r > 0, fiedler[:, 0], torch.full_like(deg_r, float("inf")))
        perm = torch.argsort(key, stable=True)
        if unfold is None:
            unfold = m == n and UNFOLD_MIN_ROWS <= n <= UNFOLD_MAX_ROWS and nnz < (1 << 31) and vectors >= UNFOLD_VECTORS
        if unfold:   # nearly degenerate leading vectors: order along the CURVE of the leading subspace (unfolded_order)
            cells = UNFOLD_CELLS
            while cells < 16384 and n > cells * UNFOLD_ROWS_PER_CELL * 4:
                cells *= 2
            lead_vectors = UNFOLD_VECTORS if cells == UNFOLD_CELLS else min(2 * UNFOLD_VECTORS, x.shape[1] // 2)
            lead = (x @ torch.from_numpy(evecs_np[:, -lead_vectors:].astype(np.float32)).to(dev)) * ir
            plain = perm
            perm, unfold_info = unfolded_order(lead, indptr, indices, deg_r > 0, cells=cells, return_info=True)
            settled = unfold_info.pop("settled")
            # trust the unfolded order only when the boxes hang together and their distance matrix IS one-dimensional (leading
            # eigenvalue of the scaling far above the second: 100-126 x on the reddit-like stand-in; 2-16 x on the products-like
            # one, whose band is 1/300 of the rows -- there the embedding is a curve only in stretches, the order comes out
            # right in parts and measured slower than it is estimated: 4.98 ms against 4.84 for the identity); otherwise the
            # plain sort stands and the caller's estimate decides as before
            unfold_info["accepted"] = bool(4 * unfold_info["ordered"] >= unfold_info["pure"]
                                           and unfold_info["one_dimensional"] >= UNFOLD_MIN_ONE_DIMENSIONAL)

            def apply_a(b):
                import warnings

                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    return spmm(*handle, num_nodes=n, num_edges=nnz, feat=b)

            if unfold_info["accepted"]:
                unfold_info["settled_rows_before_growth"] = int(settled.sum())
                perm, settled = grow_settled(apply_a, perm, settled, deg_r > 0, deg_r)
                unfold_info["settled_rows"] = int(settled.sum())
                perm = neighbour_votes(apply_a, perm, settled, deg_r > 0, degree=deg_r)
                for _ in range(max(0, vote_rounds - 1)):                                                # everybody settled: finer stretches
                    perm = neighbour_votes(apply_a, perm, deg_r > 0, deg_r > 0, buckets=512, reach=5, degree=deg_r)
            else:
                perm = plain
            stamp("unfolding")
        # ---- local refinement.  The global coordinate places a row to within the noise the graph's far (random) edges put
        # into it: every row's coordinate is the mean of its 2-hop neighbours', and on the reddit-like stand-in half of those
        # are uniformly random rows (position noise of a few thousand rows).  A row's NEAR neighbours pin it down much
        # better.  With the current positions p (ranks), take a partition of unity of `bumps` hat functions over the
        # positions, features phi_b(p_j) and phi_b(p_j) p_j, push both through K (two more products, 2 x bumps columns) and
        # give every row the mean position of its 2-hop neighbours UNDER THE BUMPS NEAR ITS OWN POSITION only -- far
        # neighbours fall under other bumps and are ignored.  Each pass shrinks the noise by about the square root of the
        # near neighbours per row.
        nz = int((deg_r > 0).sum())
        for step in range(refine if nz > 4 * refine_width else 0):
            pos = torch.empty(n, device=dev)
            pos[perm] = torch.arange(n, device=dev, dtype=torch.float32)
            # the bumps narrow from pass to pass (refine_width, /2, /4, /4 ...): a narrower window keeps fewer of the random
            # 2-hop neighbours (three quarters of all 2-hop neighbours on the reddit-like stand-in) beside the local ones
            bumps = min(124, max(2, int(round(nz / (refine_width / (1 << min(step, 2)))))))
            width = nz / bumps
            centres = (torch.arange(bumps, device=dev, dtype=torch.float32) + 0.5) * width
            phi = (1.0 - (pos[:, None] - centres[None, :]).abs() / width).clamp(min=0.0)
            phi[:, 0] = torch.where(pos < centres[0], torch.ones_like(pos), phi[:, 0])          # flat beyond the end centres
            phi[:, -1] = torch.where(pos > centres[-1], torch.ones_like(pos), phi[:, -1])
            phi = phi * (deg_r > 0)[:, None]
            pad = (-2 * bumps) % 8
            both = torch.cat([phi, phi * (pos[:, None] / nz), torch.zeros(n, pad, device=dev)], dim=1)
            pushed = apply_k(both / ir.clamp(min=1e-20)) / ir.clamp(min=1e-20)   # D_r^-1 A D_c^-1 A^T: plain 2-hop averages
            mass, moment = pushed[:, :bumps], pushed[:, bumps:2 * bumps]
            near = phi > 0                                                       # the (at most two) bumps over the row itself
            den = (mass * near).sum(1)
            new_pos = torch.where(den > 0, (moment * near).sum(1) / den.clamp(min=1e-30) * nz, pos)
            key = torch.where(deg_r > 0, new_pos, torch.full_like(deg_r, float("inf")))
            perm = torch.argsort(key, stable=True)
        stamp("local refinement")
        info = {"eigenvalues": evals[-4:].flip(0).tolist(), "iterations": iterations, "vectors": vectors, "refine": refine,
                "unfolded": unfold_info if unfold else False, "extra_rounds": extra_rounds,
                "phase_ms": {b[0]: round((b[1] - a[1]) * 1e3, 2) for a, b in zip(stamps, stamps[1:])}}
    return (perm, info) if return_info else perm


def refine_by_votes(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, perm: torch.Tensor, rounds=((256, 3), (512, 5))):
    """Neighbour votes on an order that is right at the scale of thousands of rows (the chained cluster order, round 6): per round one
    SpMM of the one-hot stretch matrix (the product's own kernel, default tile, no side-car) counts every row's neighbours per stretch
    of the current order and the row moves to the centroid of its densest window (``neighbour_votes``).  ``rounds``: (stretches, reach)."""
    from . import hybrid
    from .jit_kernels import spmm as spmm_wrapper
    from .spmm.spmm import csr_preprocess_device, spmm

    import warnings

    n = num_nodes
    deg = (indptr[1:] - indptr[:-1]).float()
    active = deg > 0
    nnz = int(indices.numel())
    with spmm_wrapper.tune_space("none"), hybrid.mode_override("0"):
        handle = csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = None

        def apply_a(b):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return spmm(*handle, num_nodes=n, num_edges=nnz, feat=b)

        for buckets, reach in rounds:
            perm = neighbour_votes(apply_a, perm, active, active, buckets=buckets, reach=reach, degree=deg)
    return perm


def degree_permutation_device(indptr: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Rows by descending degree (stable), on the input's device: equal-length windows, no locality."""
    deg = (indptr[1:] - indptr[:-1]).long()
    return torch.argsort(-deg, stable=True)


def permute_rows_csr(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, perm: torch.Tensor):
    """CSR of ``A[perm, :]`` (rows regrouped, column ids untouched), int32, on the input's device."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    pdeg = deg[perm]
    new_indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    new_indptr[1:] = torch.cumsum(pdeg, 0)
    shift = torch.repeat_interleave(indptr[:-1].long()[perm] - new_indptr[:-1], pdeg)
    new_indices = indices[shift + torch.arange(indices.numel(), device=dev)]   # every row is kept: no host read for the size
    return new_indptr.to(torch.int32), new_indices.contiguous()


def relabel_csr(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, perm: torch.Tensor):
    """CSR of ``P A P^T`` for the order ``perm`` (position k holds old node perm[k]): rows regrouped AND column ids relabelled
    with the same permutation, rows sorted -- what the reference's externally reordered ``<name>.reorder.npz`` files hold
    (bench/graph_gen.py:42-45).  int32, on the input's device; one 64-bit sort of the edge keys."""
    dev = indptr.device
    label = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    label[perm] = torch.arange(num_nodes, dtype=torch.int64, device=dev)      # old node -> new node
    p_indptr, p_indices = permute_rows_csr(indptr, indices, num_nodes, perm)
    deg = (p_indptr[1:] - p_indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_nodes, dtype=torch.int64, device=dev), deg)
    keys = torch.sort(rows * num_nodes + label[p_indices.long()]).values
    del rows
    new_indices = (keys % num_nodes).to(torch.int32)
    return p_indptr, new_indices.contiguous()


# B's address locality (round 5).  Rows of B referenced within LOCAL_RADIUS of the referencing row stay in the XCD's 4 MiB L2
# while its window range sweeps them (4 MiB / 256-byte rows = 16 k rows, shared by the range's co-resident windows); the others
# come from the Infinity Cache or HBM at about half the rate (MI355X_MICROARCH.md "Indexed rows": 66-73 vs 33.5 GB/s per CU).
# The block-count estimate below was fitted on graphs whose edges are half local (reddit-like), so a candidate's estimate is
# scaled by gather_cost(its local fraction) / gather_cost(0.5).  Only orders that RELABEL the columns can change the fraction.
LOCAL_RADIUS = 4096
LOCAL_RATE, FAR_RATE, LOCAL_REFERENCE = 70.0, 33.5, 0.5


def local_fraction(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, label: torch.Tensor = None) -> float:
    """Fraction of the edges whose column lies within LOCAL_RADIUS of its row (``label``: in the labels of a candidate order,
    old node -> new node).  One pass over the edges; one host read."""
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_nodes, dtype=torch.int64, device=indptr.device), deg)
    cols = indices.long()
    if label is not None:
        rows, cols = label[rows], label[cols]
    return float(((rows - cols).abs() <= LOCAL_RADIUS).float().mean()) if cols.numel() else 1.0


def locality_factor(local: float) -> float:
    cost = lambda x: x / LOCAL_RATE + (1.0 - x) / FAR_RATE      # noqa: E731
    return cost(local) / cost(LOCAL_REFERENCE)


# ---- method="auto": candidates judged by the format's own statistics, identity kept unless one clearly pays (round 4) ------
AUTO_MIN_GAIN = 0.03           # a candidate must cut the estimated step by this much
# ... and by this much when the columns are relabelled too (round 5): an order that is right only in stretches (products-like,
# shuffled: band 1/300 of the rows) was estimated 7 % faster than the labels it came with and measured 3 % slower
AUTO_MIN_GAIN_RELABEL = 0.10
# subspace steps of the spectral candidate: 8 give the same order quality as 16 on both reddit-size stand-ins (TC blocks 12.749 M
# / 12.749 M, k-steps 224.6 k / 224.3 k; block model 10.348 M / 10.339 M) at 68 instead of 95 ms (profiles/r04/experiment_spectral_cost.log)
AUTO_SPECTRAL_ITERATIONS = 8
# The estimate (ms, F = 128 16-bit on MI355X; only RATIOS between orders of one graph are used).  Window format: TC blocks x
# 0.14 us (14.11 M -> 1.78 ms shuffled, 12.86 M -> 1.90 natural, 11.05 M -> 1.45 block model).  Two-level: the longest of
#   (a) the residual's gathers beside the panel kernel: residual edges / 8 (one edge per gathered row: TC blocks) x 0.21 us
#       (6.36 M -> 1.30 ms, 3.76 M -> 0.89),
#   (b) the panel kernel's k-steps per CU x 1.8 us beside the window kernel (713 -> 1.28 ms, 525 -> 0.96),
#   (c) its critical path: the longest PIECE (hybrid.panel_parts cuts panels at default_part_cap) x 2 us per k-step -- without
#       pieces the breadth-first order's hub panel of 5,547 k-steps ran 5.7 ms whatever the other 454 panels did,
# plus 0.05 ms of zero fill and combine passes.  profiles/r04/experiment_reorder_auto*.log, experiment_panel_parts.log
AUTO_MS_PER_WINDOW_BLOCK = 1.4e-7
AUTO_MS_PER_RESIDUAL_BLOCK = 2.1e-7
AUTO_MS_PER_KSTEP_PER_CU = 1.8e-3
AUTO_MS_PER_PANEL_KSTEP = 2.0e-3
AUTO_MS_JOIN = 0.05
AUTO_MIN_MEAN_DEGREE = 64      # below this mean degree no candidate is tried: 16 rows x 50 edges over millions of columns share no
                               # column whatever their order (products-like: 15.44 / 15.53 / 15.52 M TC blocks natural / shuffled /
                               # reordered, profiles/HISTORY.md section 3.4), and the two searches cost 0.47 s there


AUTO_TIE_BAND = 0.25           # candidates estimated within this of the best one are measured against each other (round 6)
AUTO_TIE_FEATS = 128


def measured_step_ms(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, perm: torch.Tensor, iters: int = 5) -> float:
    """Milliseconds of one ``spmm`` (fp16 features, AUTO_TIE_FEATS columns) on the handle of ``P A P^T`` for the order ``perm`` --
    the tie-break of ``auto_permutation``: the operator's own preprocess (two-level decision included) and its own tile choice."""
    import warnings

    from .spmm.spmm import csr_preprocess_device, spmm

    p_indptr, p_indices = relabel_csr(indptr, indices, num_nodes, perm)
    handle = csr_preprocess_device(p_indptr, p_indices, num_nodes)
    handle[1].hash_tag = None
    gen = torch.Generator(device=indptr.device)
    gen.manual_seed(0)
    feat = torch.randn(num_nodes, AUTO_TIE_FEATS, device=indptr.device, generator=gen).half()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # the untagged-handle warning of feature_hash
        for _ in range(3):
            spmm(*handle, num_nodes=num_nodes, num_edges=int(indices.numel()), feat=feat)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            spmm(*handle, num_nodes=num_nodes, num_edges=int(indices.numel()), feat=feat)
        end.record()
        end.synchronize()
    return start.elapsed_time(end) / iters


def order_statistics(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None) -> dict:
    """What a row order does to the block format, from the COUNT phases of the two preprocess builders only (no handle is
    built): TC blocks of the window format, the fraction of the edges in columns that >= tau rows of a 512-row panel share,
    the panel plan's k-steps in all and in its longest panel, the residual's edges, whether ``csr_preprocess`` would build the
    two-level side-car for this order, and ``estimated_ms`` -- the step by the model above.  One host sync."""
    from . import capi, hybrid

    dev = indptr.device
    num_cols = num_nodes if num_cols is None else int(num_cols)
    stream = torch.cuda.current_stream().cuda_stream
    windows = (num_nodes + 15) // 16
    ws = torch.empty(capi.csr_preprocess_workspace_bytes(num_nodes, num_cols, indices.numel()), dtype=torch.uint8, device=dev)
    part = torch.empty(windows, dtype=torch.int32, device=dev)
    pointer1 = torch.empty(windows + 1, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    capi.launch_csr_window_count(indptr, indices, num_nodes, num_cols, ws, part, pointer1, status, stream)
    waves, rb, tau = hybrid.DEFAULT_WAVES, hybrid.DEFAULT_ROW_BLOCKS, hybrid.DEFAULT_TAU
    plan_ok = (num_cols <= hybrid.MAX_PLAN_COLS and num_nodes > 0
               and ((num_cols + (1 << 16) - 1) >> 16) * indices.numel() <= hybrid.MAX_PLAN_EDGE_PASSES)
    if plan_ok:
        panels = (num_nodes + waves * rb * 16 - 1) // (waves * rb * 16)
        pws = torch.empty(capi.panel_plan_workspace_bytes(num_nodes, waves, rb), dtype=torch.uint8, device=dev)
        panel_ptr = torch.empty(panels + 1, dtype=torch.int32, device=dev)
        resid_ptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        pstatus = torch.empty(1, dtype=torch.int32, device=dev)
        capi.check(capi.launch_panel_plan_count(indptr, indices, num_nodes, num_cols, waves, rb, tau, pws, panel_ptr,
                                                resid_ptr, pstatus, stream), "voltrix_launch_panel_plan_count")
        longest = (panel_ptr[1:] - panel_ptr[:-1]).max()[None]
        blocks, outside, ksteps, resid, bad, longest = torch.cat(
            [pointer1[-1:], status, panel_ptr[-1:], resid_ptr[-1:], pstatus, longest.to(torch.int32)]).tolist()
        plan_ok = bad == 0
    else:
        blocks, outside = torch.cat([pointer1[-1:], status]).tolist()
        ksteps = resid = longest = 0
    nnz = int(indices.numel())
    share = (nnz - resid) / max(1, nnz) if plan_ok else 0.0
    big = (nnz >= hybrid.AUTO_MIN_EDGES and num_nodes >= hybrid.AUTO_MIN_ROWS
           and nnz >= hybrid.AUTO_MIN_MEAN_DEGREE * max(1, num_nodes))
    mode = hybrid.hybrid_mode()       # the decision csr_preprocess_device will make for this order (spmm/spmm.py)
    two_level = (plan_ok and ksteps > 0 and (mode == "on" or (mode in ("auto", "tune") and big))
                 and share >= hybrid.min_shared_fraction())
    if two_level:
        piece = min(int(longest), hybrid.default_part_cap(int(ksteps))) if hybrid.PANEL_PART_FACTOR > 0 else int(longest)
        estimate = AUTO_MS_JOIN + max(AUTO_MS_PER_RESIDUAL_BLOCK * resid / 8.0,
                                      AUTO_MS_PER_KSTEP_PER_CU * ksteps / hybrid.NUM_CUS, AUTO_MS_PER_PANEL_KSTEP * piece)
    else:
        estimate = AUTO_MS_PER_WINDOW_BLOCK * float(blocks)
    return {"tc_blocks": int(blocks), "shared_fraction": share, "ksteps": int(ksteps), "longest_panel_ksteps": int(longest),
            "two_level": bool(two_level), "ids_outside_universe": int(outside), "residual_edges": int(resid) if plan_ok else nnz,
            "estimated_ms": estimate}


def auto_permutation(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                     candidates=("bfs", "spectral", "clusters", "clusters+votes"), info: dict = None, relabel: bool = False):
    """The safe default: every candidate order is judged by ``order_statistics`` of the row-permuted CSR
    against the order the caller's rows already have, and the IDENTITY is kept unless a candidate cuts the ESTIMATED step by
    ``AUTO_MIN_GAIN`` (3 %).  The estimate is the longer of the gather volume (fewer TC blocks, more shared edges) and the panel
    kernel's critical path: a degenerate breadth-first order (its levels turn into a degree sort on graphs with a background of
    random edges) piles the shared columns of 512 hub rows into one panel whose workgroup then runs alone -- 5.7 ms against
    1.78 ms on the label-shuffled reddit-like graph although it has FEWER TC blocks and MORE shared edges (DESIGN.md section
    3.4).  Deterministic: statistics of the format, no timing.  Returns ``(perm or None, name)``; None = keep the caller's
    order.  ``relabel`` (round 5): the candidates are judged as SYMMETRIC relabellings ``P A P^T`` -- the estimate of every
    order, the caller's included, is scaled by what its share of local edges does to the gather rate (``locality_factor``), the
    one effect a row-only order cannot have -- and the mean-degree guard is lifted: at low degree the order does not change the
    block count, but where B's rows sit in memory is all that is left to win."""
    report = {"identity": order_statistics(indptr, indices, num_nodes, num_cols)}
    base = report["identity"]
    if relabel:
        base["local_fraction"] = local_fraction(indptr, indices, num_nodes)
        base["estimated_ms"] *= locality_factor(base["local_fraction"])
    best, best_name, best_ms = None, "identity", base["estimated_ms"]
    runners = []                # (estimate, name, perm) of every candidate that clears the gain threshold
    cluster_perm = None
    if indices.numel() < AUTO_MIN_MEAN_DEGREE * max(1, num_nodes) and not relabel:
        candidates = ()
    for name in candidates:
        if name == "bfs":
            perm = bfs_permutation(indptr, indices, num_nodes, num_cols)
        elif name == "spectral":
            if num_nodes < 4 * 8192:        # a one-dimensional embedding of a few thousand rows regroups nothing a window sees
                continue
            perm = spectral_permutation(indptr, indices, num_nodes, num_cols, iterations=AUTO_SPECTRAL_ITERATIONS)
        elif name == "degree":
            perm = degree_permutation_device(indptr, num_nodes)
        elif name == "clusters":
            from . import cluster_order

            if not relabel or 2 * indices.numel() > cluster_order.MAX_EDGES or num_cols not in (None, num_nodes):
                continue
            cluster_info = {}
            perm = cluster_order.cluster_permutation(indptr, indices, num_nodes, info=cluster_info)
            cluster_perm = perm
        elif name == "clusters+votes":
            # the chained cluster order is right at the scale of thousands of rows; two rounds of neighbour votes (one SpMM each) settle
            # the rows inside it: co-purchase stand-ins, shuffled: 1.06 / 1.02 / 1.02 -> 0.98 / 0.96 / 0.95 x the generating order's step
            # (profiles/r06/experiment_reorder_cluster_votes.log); needs the candidate before it
            if cluster_perm is None or not indptr.is_cuda:
                continue
            cluster_info = None
            perm = refine_by_votes(indptr, indices, num_nodes, cluster_perm)
        else:
            raise ValueError(f"unknown candidate order {name!r}")
        p_indptr, p_indices = permute_rows_csr(indptr, indices, num_nodes, perm)
        st = order_statistics(p_indptr, p_indices, num_nodes, num_cols)     # block counts do not depend on the column labels
        del p_indptr, p_indices
        if relabel:
            label = torch.empty(num_nodes, dtype=torch.int64, device=indptr.device)
            label[perm] = torch.arange(num_nodes, dtype=torch.int64, device=indptr.device)
            st["local_fraction"] = local_fraction(indptr, indices, num_nodes, label)
            st["estimated_ms"] *= locality_factor(st["local_fraction"])
            del label
        eligible = bool(st["estimated_ms"] <= (1.0 - (AUTO_MIN_GAIN_RELABEL if relabel else AUTO_MIN_GAIN)) * base["estimated_ms"])
        st["accepted"] = bool(eligible and st["estimated_ms"] < best_ms)
        if name == "clusters" and cluster_info is not None:
            st["clusters"] = cluster_info
        report[name] = st
        if eligible:
            runners.append((st["estimated_ms"], name, perm))
        if st["accepted"]:
            best, best_name, best_ms = perm, name, st["estimated_ms"]
    # round 6: candidates whose estimates are within AUTO_TIE_BAND of the best one are told apart by MEASUREMENT (relabelled
    # handles on the device only): the statistics cannot see what separates two orders of the same block counts and the same
    # share of local edges -- protein-like, shuffled: spectral 1.754 / clusters 1.817 estimated, 0.920 / 0.832 ms measured
    close = [r for r in runners if r[0] <= (1.0 + AUTO_TIE_BAND) * best_ms]
    if relabel and indptr.is_cuda and len(close) >= 2:
        timed = {}
        for _, name, perm in close:
            timed[name] = measured_step_ms(indptr, indices, num_nodes, perm)
            report[name]["measured_ms"] = timed[name]
        best_name = min(timed, key=timed.get)
        best = next(r[2] for r in close if r[1] == best_name)
        for name in report:
            if name != "identity":
                report[name]["accepted"] = name == best_name
    if info is not None:
        info.update(report=report, picked=best_name)
    return best, best_name


@dataclasses.dataclass(eq=False)
class ReorderedHandle:
    """Reference-format handle of ``A[perm, :]`` + the map that sends its rows back: not a tuple, because the three
    tensors alone describe the row-permuted matrix."""
    blk_offsets: torch.Tensor
    hspa_packed: torch.Tensor
    hind: torch.Tensor
    row_map: torch.Tensor        # int32 [16 W]: handle row -> row of C, -1 for the padding rows of the last window; None = identity
    perm: torch.Tensor           # int64 [N]
    num_nodes: int
    num_edges: int
    method: str
    relabelled: bool = False     # True: the handle is that of P A P^T (columns relabelled too): B and C live in the NEW order
    # round 6 -- separable edge values v_ij = r_i c_j (normalised adjacencies) on a reordered handle: the binary product between two row
    # scalings, as in voltrix/weighted.py.  Stored in the orders spmm_reordered works in: col_scale in the order of the B it takes,
    # row_scale in the order of the C it returns without ``unpermute`` (both the NEW order on relabelled handles, the caller's otherwise)
    row_scale: torch.Tensor = None
    col_scale: torch.Tensor = None


def csr_preprocess_reordered(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                             method="auto", info: dict = None, relabel: bool = False, row_scale: torch.Tensor = None,
                             col_scale: torch.Tensor = None) -> ReorderedHandle:
    """CSR (CPU or CUDA int32) -> handle of the row-reordered matrix for ``spmm_reordered``.  ``method``: "auto" (default,
    round 4: the breadth-first and the spectral order are tried and judged by the format's own statistics; the caller's order
    is KEPT unless one of them clearly pays -- never worse than no reorder, ``auto_permutation``), "bfs" (Cuthill-McKee
    levels), "clusters" (round 6: connected components + multilevel size-constrained label propagation, voltrix/cluster_order.py --
    the order for graphs of mean degree 2-12, where neither a search nor an eigenvector survives the random edges; tried by
    "auto" for relabelled handles), "spectral" (Fiedler order of the row co-occurrence matrix, computed with the SpMM kernels: the one that survives a
    background of random edges), "degree", "identity", or an explicit permutation tensor (position k holds row perm[k]).
    ``info``: dict that receives the statistics ``auto`` decided on.

    ``relabel=True`` (round 5; square adjacency): the SYMMETRIC form ``P A P^T`` -- the columns are relabelled with the row
    permutation, exactly what the reference benchmarks on (``<name>.reorder.npz``, bench/graph_gen.py:42-45,
    bench_all.py:120-129).  The handle then describes the relabelled matrix: ``spmm_reordered`` takes B in the NEW order
    (``permute_features(handle, feat)``, once per feature matrix) and returns C in the new order (``unpermute=True`` puts the
    rows back).  Only this form restores the address locality of B (a row-only order cannot: products-like 4.4 ms natural, 4.9
    shuffled, 5.7 row-reordered); ``auto`` judges the candidates with that term and tries them at any mean degree.

    ``row_scale`` / ``col_scale`` (round 6; float [N] / [num_cols], in the CALLER's node order): edge values ``v_ij = r_i c_j`` -- the
    normalised adjacency a GCN layer multiplies by -- on the reordered handle: ``spmm_reordered`` computes
    ``diag(r) A (diag(c) B)`` on the binary operator between two ``scale_rows`` passes (voltrix/weighted.py).  The vectors are
    permuted with the handle; factors of general values: ``weighted.separable_scales``."""
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_nodes + 1
    assert not relabel or num_cols in (None, num_nodes), "relabel=True relabels rows and columns alike: a square adjacency"
    indptr_d, indices_d = indptr.contiguous().cuda(), indices.contiguous().cuda()
    if method == "auto":
        perm, name = auto_permutation(indptr_d, indices_d, num_nodes, num_cols, info=info, relabel=relabel)
        if perm is None:
            perm, name = torch.arange(num_nodes, device=indptr_d.device), "auto:identity"
        else:
            name = "auto:" + name
    elif method == "identity":
        perm, name = torch.arange(num_nodes, device=indptr_d.device), "identity"
    elif isinstance(method, torch.Tensor):
        perm, name = method.to(indptr_d.device, torch.int64), "given"
        assert perm.numel() == num_nodes and int(torch.sort(perm).values.ne(torch.arange(num_nodes, device=perm.device)).sum()) == 0
    elif method == "bfs":
        perm, name = bfs_permutation(indptr_d, indices_d, num_nodes, num_cols), "bfs"
    elif method == "spectral":
        perm, name = spectral_permutation(indptr_d, indices_d, num_nodes, num_cols), "spectral"
    elif method == "degree":
        perm, name = degree_permutation_device(indptr_d, num_nodes), "degree"
    elif method in ("clusters", "clusters+votes"):
        from . import cluster_order

        perm, name = cluster_order.cluster_permutation(indptr_d, indices_d, num_nodes, info=info), method
        if method == "clusters+votes":
            perm = refine_by_votes(indptr_d, indices_d, num_nodes, perm)
    else:
        raise ValueError(f"unknown reorder method {method!r}")
    if name.endswith("identity"):
        p_indptr, p_indices = indptr_d, indices_d
    elif relabel:
        p_indptr, p_indices = relabel_csr(indptr_d, indices_d, num_nodes, perm)
    else:
        p_indptr, p_indices = permute_rows_csr(indptr_d, indices_d, num_nodes, perm)
    # the operator's own preprocess: the reference handle of A[perm, :] and, when the (reordered!) graph pays for it, the
    # two-level side-car -- panels are 512 consecutive rows of the NEW order, which is where the reorder put the rows that
    # share columns
    from .spmm.spmm import csr_preprocess_device

    pointer1, hspa_packed, hind = csr_preprocess_device(p_indptr, p_indices, num_nodes, num_cols)
    padded = 16 * ((num_nodes + 15) // 16)
    row_map = None
    if not name.endswith("identity") and not relabel:   # the identity order and relabelled handles need no map: C is written in place
        row_map = torch.full((padded,), -1, dtype=torch.int32, device=indptr_d.device)
        row_map[:num_nodes] = perm.to(torch.int32)
    handle = ReorderedHandle(pointer1, hspa_packed, hind, row_map, perm, num_nodes, int(indices.numel()), name,
                             relabelled=bool(relabel) and not name.endswith("identity"))
    if row_scale is not None:
        assert row_scale.numel() == num_nodes
        r = row_scale.to(indptr_d.device, torch.float32)
        handle.row_scale = (r.index_select(0, perm) if handle.relabelled else r).contiguous()
    if col_scale is not None:
        assert col_scale.numel() == (num_nodes if num_cols is None else num_cols)
        c = col_scale.to(indptr_d.device, torch.float32)
        handle.col_scale = (c.index_select(0, perm) if handle.relabelled else c).contiguous()
    return handle


def permute_features(handle: ReorderedHandle, feat: torch.Tensor) -> torch.Tensor:
    """``feat`` (rows in the caller's node order) -> the rows in the order of a relabelled handle: row k = old row perm[k].  Once
    per feature matrix, outside the product (the reference's protocol reads features for the reordered file the same way)."""
    assert isinstance(handle, ReorderedHandle) and feat.shape[0] == handle.num_nodes
    return feat.index_select(0, handle.perm.to(feat.device)) if handle.relabelled else feat


def unpermute_output(handle: ReorderedHandle, out: torch.Tensor) -> torch.Tensor:
    """C of a relabelled handle (rows in the new order) -> rows in the caller's node order."""
    return torch.empty_like(out).index_copy_(0, handle.perm.to(out.device), out) if handle.relabelled else out


def spmm_reordered(handle: ReorderedHandle, feat: torch.Tensor, hash_tag: str = None, unpermute: bool = False) -> torch.Tensor:
    """``csr(ones) @ feat`` for a ``ReorderedHandle``: float32 [num_nodes, F] on the current stream.
    Row-only handles (``relabel=False``): ``feat`` as for ``voltrix.spmm`` (it is gathered as it is: column ids were never
    relabelled), the result in the ORIGINAL row order.  Relabelled handles: ``feat`` in the NEW order (``permute_features``),
    the result in the new order too -- the plain operator on ``P A P^T``, every kernel and format of it -- unless
    ``unpermute``."""
    from .jit_kernels import spmm_kernel
    from .spmm.spmm import _operand

    assert isinstance(handle, ReorderedHandle)
    if hash_tag is not None and getattr(handle.hspa_packed, "hash_tag", None) is None:
        handle.hspa_packed.hash_tag = hash_tag
    if handle.row_scale is not None or handle.col_scale is not None:      # separable edge values: two row scalings around the binary product
        from .weighted import scale_rows_of

        scaled = feat if handle.col_scale is None else scale_rows_of(feat, handle.col_scale)
        binary = dataclasses.replace(handle, row_scale=None, col_scale=None)
        out = spmm_reordered(binary, scaled, unpermute=False)
        if handle.row_scale is not None:
            out = scale_rows_of(out, handle.row_scale, in_place=True)
        return unpermute_output(handle, out) if (unpermute and handle.relabelled) else out
    num_feats = feat.shape[1]
    from .spmm.spmm import spmm, two_level_of

    if handle.row_map is None:       # identity order (method "auto" kept the caller's rows) or a relabelled handle: the plain operator
        out = spmm(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                   num_edges=handle.num_edges, feat=feat)
        return unpermute_output(handle, out) if (unpermute and handle.relabelled) else out
    if two_level_of(handle.hspa_packed) is not None:
        # two-level side-car on the reordered rows: the panel kernel writes its panel's rows in place, so the product comes
        # out in the handle's row order and one indexed copy (N x F x 4 bytes each way) puts the rows back
        permuted = spmm(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                        num_edges=handle.num_edges, feat=feat)
        return torch.empty_like(permuted).index_copy_(0, handle.perm, permuted)
    # fp32 features: the same decision voltrix.spmm makes for this handle (exact fp32 tiles on handles of short windows, else
    # the scaled fp16 cast) -- one handle, one numerics, whichever entry point (round 6)
    from .spmm.spmm import fp32_mode

    operand, out_scale, padded, _ = _operand(feat, fp32_mode(handle.hspa_packed, handle.num_nodes, num_feats)
                                             if feat.dtype == torch.float32 else None)
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    spmm_kernel(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                num_edges=handle.num_edges, embedding_dim=padded, input=operand, output=output, out_scale=out_scale,
                row_map=handle.row_map)
    return output if padded == num_feats else output[:, :num_feats].contiguous()
