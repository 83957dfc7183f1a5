"""Locality reorder of a graph before ``csr_preprocess`` (the "reorder" half).

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
