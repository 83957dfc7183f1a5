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


def round_fp16_scaled(feat: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 mantissa with one power-of-two scale per tensor, e = exponent(max |x|) - 14 (clamped to >= -100;
    0 if the tensor is all zero or holds Inf / NaN): what ``voltrix.spmm`` does to an fp32 operand
    (include/voltrix_capi.h, voltrix_launch_cast_f32_f16_scaled).  Result in an fp32 container."""
    feat = torch.as_tensor(feat, dtype=torch.float32)
    amax = feat.abs().max() if feat.numel() else torch.tensor(0.0)
    if not torch.isfinite(amax) or amax == 0:
        e = 0
    else:
        e = max(int(torch.frexp(amax)[1]) - 1 - 14, -100)   # frexp: amax = m * 2^exp with m in [0.5, 1)
    return torch.ldexp(torch.ldexp(feat, torch.tensor(-e)).to(torch.float16).to(torch.float32), torch.tensor(e))


def spmm(indptr, indices, feat, num_rows, operand_rounding=None):
    """fp32 CPU SpMM.  ``operand_rounding="fp16"`` evaluates the oracle on ``feat.half().float()``
    so that only accumulation-order error remains; ``"fp16-scaled"`` on
    :func:`round_fp16_scaled`."""
    feat = torch.as_tensor(feat)
    if operand_rounding == "fp16-scaled":
        feat = round_fp16_scaled(feat)
    elif operand_rounding == "fp16":
        feat = feat.to(torch.float16)
    elif operand_rounding == "bf16":
        feat = feat.to(torch.bfloat16)
    feat = feat.to(torch.float32)
    return csr_ones(indptr, indices, num_rows, feat.shape[0]) @ feat
