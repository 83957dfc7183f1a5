"""Backward pass of the SpMM: ``dB = A^T @ dC`` through a transposed handle.

No reference counterpart -- the reference is forward-only (voltrix/jit_kernels/spmm.py:53-54 asserts a plain float
input and nothing registers a gradient), which is what keeps it out of GCN *training* loops.  ``A`` is binary, so the
gradient with respect to the dense operand of ``C = A @ B`` is the SpMM of the transposed adjacency with the incoming
gradient; the same kernels run it on a second handle built from the transposed CSR (one device-side sort by column).

    op = voltrix.autograd.SpMM(indptr, indices, num_nodes)        # CPU or CUDA int32 CSR; builds A and A^T handles
    out = op(feat)                                                # float32 [num_nodes, F]; feat.requires_grad honoured
    out.sum().backward()                                          # feat.grad = A^T @ 1

The forward keeps ``voltrix.spmm``'s numerics (fp16 / bf16 operand or scaled-fp16 rounding of an fp32 operand, fp32
accumulate); the backward treats the incoming gradient the same way, i.e. the pair is the exact adjoint up to the
operand rounding the forward applies too.
"""
from __future__ import annotations

import torch

from .spmm.spmm import csr_preprocess_device, spmm


def csr_transpose_device(indptr: torch.Tensor, indices: torch.Tensor, num_rows: int, num_cols: int):
    """CSR of ``A^T`` ([num_cols, num_rows]) for a device CSR of ``A`` ([num_rows, num_cols]): int32, rows sorted,
    duplicates kept (they count once in the block format, like everywhere else)."""
    assert indptr.is_cuda and indices.is_cuda and indptr.dtype == torch.int32 and indices.dtype == torch.int32
    from . import capi

    # row ids expanded per entry + one stable radix sort by column + row pointers by binary search (reorder_kernels.hpp;
    # round 2 sorted 64-bit (col, row) keys with torch ops)
    return capi.csr_transpose(indptr.contiguous(), indices.contiguous(), num_rows, num_cols)


class _SpMMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, op):
        ctx.op = op
        ctx.in_dtype = feat.dtype
        if op.weighted is not None:
            from .weighted import spmm_weighted

            return spmm_weighted(op.weighted, feat)
        return spmm(*op.handle, num_nodes=op.num_rows, num_edges=op.num_edges, feat=feat)

    @staticmethod
    def backward(ctx, grad_out):
        op = ctx.op
        if op.weighted is not None:
            from .weighted import spmm_weighted

            return spmm_weighted(op.weighted_t, grad_out.contiguous()).to(ctx.in_dtype), None
        grad = spmm(*op.handle_t, num_nodes=op.num_cols, num_edges=op.num_edges, feat=grad_out.contiguous())
        return grad.to(ctx.in_dtype), None


class SpMM:
    """``C = A @ B`` with a gradient for ``B``.  ``A``: binary CSR [num_rows, num_cols] (``num_cols`` defaults to
    ``num_rows``), or, with ``values`` (round 6), the weighted matrix ``csr(values)``.  Builds two reference-format handles on the current device (A and A^T); both go through
    ``voltrix.spmm`` -- tuner, schedules and the two-level side-car included."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, num_rows: int, num_cols: int = None, hash_tag: str = None,
                 values: torch.Tensor = None):
        assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_rows + 1
        self.num_rows = num_rows
        self.num_cols = num_rows if num_cols is None else int(num_cols)
        self.num_edges = int(indices.numel())
        indptr_d, indices_d = indptr.contiguous().cuda(), indices.contiguous().cuda()
        self.weighted = self.weighted_t = None
        if values is not None:
            # round 6: edge values.  Separable ones (v_ij = r_i c_j: normalised adjacencies) run both directions on the binary
            # operator between two row scalings -- A^T has the factors swapped --; general ones through value planes of A and A^T
            from .weighted import WeightedHandle, csr_preprocess_weighted, transpose_weighted

            values_d = values.contiguous().cuda()
            self.weighted = csr_preprocess_weighted(indptr_d, indices_d, values_d, num_rows, num_cols=self.num_cols)
            if self.weighted.separable:
                t_indptr, t_indices = csr_transpose_device(indptr_d, indices_d, num_rows, self.num_cols)
                t_handle = csr_preprocess_device(t_indptr, t_indices, self.num_cols, num_cols=num_rows)
                self.weighted_t = WeightedHandle(*t_handle, None, self.num_cols, self.num_edges,
                                                 row_scale=self.weighted.col_scale, col_scale=self.weighted.row_scale)
            else:
                t_indptr, t_indices, t_values = transpose_weighted(indptr_d, indices_d, values_d, num_rows, self.num_cols)
                self.weighted_t = csr_preprocess_weighted(t_indptr, t_indices, t_values, self.num_cols, num_cols=num_rows,
                                                          separable=False)
            self.handle = (self.weighted.blk_offsets, self.weighted.hspa_packed, self.weighted.hind)
            self.handle_t = (self.weighted_t.blk_offsets, self.weighted_t.hspa_packed, self.weighted_t.hind)
            if hash_tag is not None:
                self.handle[1].hash_tag = hash_tag
                self.handle_t[1].hash_tag = hash_tag + "/transposed"
            return
        self.handle = csr_preprocess_device(indptr_d, indices_d, num_rows, num_cols=self.num_cols)
        t_indptr, t_indices = csr_transpose_device(indptr_d, indices_d, num_rows, self.num_cols)
        self.handle_t = csr_preprocess_device(t_indptr, t_indices, self.num_cols, num_cols=num_rows)
        if hash_tag is not None:
            self.handle[1].hash_tag = hash_tag
            self.handle_t[1].hash_tag = hash_tag + "/transposed"

    def update_values(self, values: torch.Tensor) -> None:
        """New edge values on the same pattern (CSR order of the constructor's ``indices``): both directions, in place
        (``weighted.update_values``: one scatter per plane, no rebuild).  No gradient flows to ``values`` -- that is a sampled
        dense-dense product, not on this path."""
        from .weighted import csr_preprocess_weighted, transpose_order, transpose_weighted, update_values

        assert self.weighted is not None, "this operator was built without values"
        values_d = values.contiguous().cuda()
        was_separable = self.weighted.separable
        update_values(self.weighted, values_d)
        indptr_d, indices_d = self.weighted.csr[0], self.weighted.csr[1]
        if self.weighted.separable:                    # A^T has the factors swapped
            self.weighted_t.row_scale, self.weighted_t.col_scale = self.weighted.col_scale, self.weighted.row_scale
        elif was_separable:                            # the new values do not factor: A^T needs a value plane of its own now
            t_indptr, t_indices, t_values = transpose_weighted(indptr_d, indices_d, values_d, self.num_rows, self.num_cols)
            tag = getattr(self.handle_t[1], "hash_tag", None)
            self.weighted_t = csr_preprocess_weighted(t_indptr, t_indices, t_values, self.num_cols, num_cols=self.num_rows,
                                                      separable=False)
            self.handle_t = (self.weighted_t.blk_offsets, self.weighted_t.hspa_packed, self.weighted_t.hind)
            if tag is not None:
                self.handle_t[1].hash_tag = tag
        else:
            if getattr(self, "_t_order", None) is None:
                self._t_order = transpose_order(indptr_d, indices_d, self.num_rows)
            update_values(self.weighted_t, values_d[self._t_order])

    def __call__(self, feat: torch.Tensor) -> torch.Tensor:
        assert feat.shape[0] == self.num_cols
        return _SpMMFunction.apply(feat, self)
