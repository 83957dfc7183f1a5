"""GPU: the row-sharded operator on one GPU (world size 1, HIP path end to end).  The N > 1 plumbing is covered by
tests/test_dist_gloo.py on CPU; the driver runs bench.py --gpus N for the real multi-GPU numbers."""
import numpy as np
import pytest
import torch

import synth_graphs
from oracle import torch_ref

pytestmark = pytest.mark.gpu


def test_row_sharded_world1_hip_path(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.dist import RowShardedSpMM

    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.01)
    n = indptr.numel() - 1
    feat = torch.randn(n, 64).half()
    op = RowShardedSpMM(indptr, indices, n, hash_tag="dist_test")
    out = op(feat.cuda())
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert out.is_cuda and out.shape == (n, 64)
    assert float((out.cpu() - ref).norm() / ref.norm()) < 1e-6


def test_row_sharded_operator_with_two_level_format(cuda_device, monkeypatch):
    """VOLTRIX_HYBRID=1 switches the sharded operator's local handles to the two-level format (column ids index the
    gathered buffer: num_cols = world * rows_padded)."""
    import numpy as np

    import synth_graphs
    from oracle import torch_ref
    from voltrix import dist as vdist

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.006)
    n = indptr.numel() - 1
    op = vdist.RowShardedSpMM(indptr, indices, n, device=cuda_device, hash_tag="dist_two_level")
    assert getattr(op.handle[1], "panel_plan", None) is not None and op.handle[1].panel_plan.num_ksteps > 0
    feat = torch.randn(n, 64).half()
    out = op(feat.cuda())
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert float((out.cpu() - ref).norm() / ref.norm()) < 1e-5
