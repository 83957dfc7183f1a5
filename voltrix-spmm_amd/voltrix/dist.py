"""Row-window sharding across the GPUs of one node + all-gather of the dense operand (RCCL over xGMI).

No reference counterpart -- the reference is single-GPU (SURVEY.md section 2.1 "Parallelism", section 8e).
Output row windows are independent, so the sparse matrix shards by contiguous row-window ranges (balanced by
edge count, boundaries on multiples of 16 rows); every rank keeps global column ids, owns the matching row
slice of the dense operand B and produces the matching row slice of C.  The one exchange step per SpMM is an
all-gather of B.

Layout trick: ``all_gather_into_tensor`` needs equal contributions, and shards differ by up to one window plus
balance slack.  Each rank therefore contributes ``rows_padded = max shard rows`` rows and the *column ids of the
local CSR are remapped once at setup* to ``owner(col) * rows_padded + (col - row_start[owner(col)])``, so the
gathered buffer ``[world * rows_padded, F]`` is used as B directly -- no unpadding copy on the hot path.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

BLK_H = 16


def partition_rows(indptr: torch.Tensor, num_nodes: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous row ranges ``[(row_start, row_end)] * world_size``: starts are multiples of 16, edge counts as
    equal as the window granularity allows (the unit of SpMM work is the gathered row, i.e. ~ an edge).
    Deterministic: every rank computes the same partition from the same ``indptr``."""
    assert indptr.numel() == num_nodes + 1
    num_windows = (num_nodes + BLK_H - 1) // BLK_H
    ip = indptr.to(torch.int64).cpu()
    win_start = torch.arange(0, num_windows + 1, dtype=torch.int64) * BLK_H
    win_start[-1] = num_nodes
    edges_before = ip[win_start]                      # edges before each window boundary
    total = int(edges_before[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        w = int(torch.searchsorted(edges_before, torch.tensor(target, dtype=torch.int64), right=False))
        w = max(bounds[-1], min(w, num_windows))
        bounds.append(w)
    bounds.append(num_windows)
    return [(min(bounds[r] * BLK_H, num_nodes), min(bounds[r + 1] * BLK_H, num_nodes)) for r in range(world_size)]


def shard_csr(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, parts: List[Tuple[int, int]], rank: int):
    """Local CSR of ``rank`` (rows of its range, global column ids), int32, on the input's device."""
    r0, r1 = parts[rank]
    e0, e1 = int(indptr[r0]), int(indptr[r1])
    local_indptr = (indptr[r0:r1 + 1] - indptr[r0]).to(torch.int32).contiguous()
    local_indices = indices[e0:e1].to(torch.int32).contiguous()
    return local_indptr, local_indices


def remap_columns(indices: torch.Tensor, parts: List[Tuple[int, int]], rows_padded: int) -> torch.Tensor:
    """Global column id -> row of the padded all-gather buffer (see module docstring)."""
    starts = torch.tensor([p[0] for p in parts], dtype=torch.int64, device=indices.device)
    cols = indices.to(torch.int64)
    owner = torch.searchsorted(starts, cols, right=True) - 1
    return (owner * rows_padded + (cols - starts[owner])).to(torch.int32)


class RowShardedSpMM:
    """``C_local = A[rows of this rank, :] @ all_gather(B_local)`` for a binary CSR ``A``.

    ``local_preprocess(indptr_cpu_i32, indices_cpu_i32, n_rows) -> handle`` and
    ``local_spmm(handle, n_rows, n_edges, feat) -> out`` default to the HIP path (``voltrix.csr_preprocess`` /
    ``voltrix.spmm``); the CPU tests inject their own.
    """

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, group=None,
                 device: Optional[torch.device] = None, local_preprocess: Optional[Callable] = None,
                 local_spmm: Optional[Callable] = None, hash_tag: Optional[str] = None):
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_nodes = num_nodes
        self.parts = partition_rows(indptr, num_nodes, self.world_size)
        self.row_start, self.row_end = self.parts[self.rank]
        self.local_rows = self.row_end - self.row_start
        self.rows_padded = max(1, max(p[1] - p[0] for p in self.parts))
        self.device = device

        local_indptr, local_indices = shard_csr(indptr, indices, num_nodes, self.parts, self.rank)
        local_indices = remap_columns(local_indices, self.parts, self.rows_padded)
        self.local_edges = int(local_indices.numel())
        if local_preprocess is None:
            from .spmm import csr_preprocess  # HIP path; raises if the extension is missing
            gathered_rows = self.world_size * self.rows_padded  # remapped ids index the all-gather buffer

            def local_preprocess(ip, ix, n_rows):
                return csr_preprocess(ip, ix, n_rows, num_cols=gathered_rows)
        if local_spmm is None:
            from .spmm import spmm as _spmm

            def local_spmm(handle, n_rows, n_edges, feat):
                return _spmm(handle[0], handle[1], handle[2], n_rows, n_edges, feat)
        self._local_spmm = local_spmm
        self.handle = local_preprocess(local_indptr.cpu(), local_indices.cpu(), self.local_rows)
        if hash_tag is not None and hasattr(self.handle[1], "data_ptr"):
            try:
                self.handle[1].hash_tag = f"{hash_tag}_r{self.rank}of{self.world_size}"
            except AttributeError:
                pass
        self._gathered = None

    def gather(self, feat_local: torch.Tensor) -> torch.Tensor:
        """All-gather of B: ``[local_rows, F]`` per rank -> ``[world * rows_padded, F]`` (padding rows are never
        referenced by the remapped column ids)."""
        assert feat_local.dim() == 2 and feat_local.shape[0] == self.local_rows
        num_feats = feat_local.shape[1]
        if self.world_size == 1:
            return feat_local.contiguous()
        shape = (self.world_size * self.rows_padded, num_feats)
        if self._gathered is None or self._gathered.shape != shape or self._gathered.dtype != feat_local.dtype:
            self._gathered = torch.zeros(shape, dtype=feat_local.dtype, device=feat_local.device)
        if self.local_rows == self.rows_padded:
            send = feat_local.contiguous()
        else:
            send = self._gathered[self.rank * self.rows_padded:(self.rank + 1) * self.rows_padded]
            send[: self.local_rows].copy_(feat_local)
        if dist.get_backend(self.group) != "nccl":
            send = send.clone()  # only NCCL/RCCL defines the in-place (send == recv + rank * count) form
        dist.all_gather_into_tensor(self._gathered, send, group=self.group)
        return self._gathered

    def __call__(self, feat_local: torch.Tensor) -> torch.Tensor:
        full = self.gather(feat_local)
        return self._local_spmm(self.handle, self.local_rows, self.local_edges, full)
