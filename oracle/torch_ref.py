"""TEST INFRASTRUCTURE ONLY -- the reference's own numerical oracle call, on CPU:
``torch.sparse_csr_tensor(indptr_i32, indices_i32, ones_f32, (N, N)) @ feat_f32``
(tests/test_spmm.py:24-29,79-80 minus ``.cuda()``).  Also the ``cpu_baseline`` that
bench.py times on the GPU box's host cores."""
from __future__ import annotations

import torch


def csr_ones(indptr, indices, num_rows, num_cols=None):
    indptr = torch.as_tensor(indptr, dtype=torch.int32)
    indices = torch.as_tensor(indices, dtype=torch.int32)
    num_cols = num_rows if num_cols is None else num_cols
    return torch.sparse_csr_tensor(indptr, indices, torch.ones(indices.numel(), dtype=torch.float32),
                                   size=(num_rows, num_cols))


def spmm(indptr, indices, feat, num_rows, operand_rounding=None):
    """fp32 CPU SpMM.  ``operand_rounding="fp16"`` evaluates the oracle on ``feat.half().float()``
    so that only accumulation-order error remains."""
    feat = torch.as_tensor(feat)
    if operand_rounding == "fp16":
        feat = feat.to(torch.float16)
    elif operand_rounding == "bf16":
        feat = feat.to(torch.bfloat16)
    feat = feat.to(torch.float32)
    return csr_ones(indptr, indices, num_rows, feat.shape[0]) @ feat
