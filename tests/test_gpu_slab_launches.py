"""GPU: wide operands as one launch per column slab (spmm_kernels.hpp::slab_launches, round 3) against one grid over all
slabs -- same bits for the window format and for the two-level pair, and both equal to the dense product."""
import os
import subprocess
import sys

import pytest
import torch

import synth_graphs
from conftest import REPO

pytestmark = pytest.mark.gpu
WORKER = os.path.join(REPO, "tests", "slab_launch_worker.py")


def test_one_launch_per_slab_gives_the_bits_of_the_single_grid(tmp_path):
    outs = {}
    for mode in ("0", "1"):
        path = tmp_path / f"out_{mode}.pt"
        run = subprocess.run([sys.executable, WORKER, str(path)], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, VOLTRIX_SLAB_LAUNCHES=mode))
        assert run.returncode == 0, run.stderr[-3000:]
        outs[mode] = torch.load(path)
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.04)
    n = indptr.numel() - 1
    torch.manual_seed(0)
    feat = torch.randint(-4, 5, (n, 320)).float()
    want = torch.sparse_csr_tensor(indptr.long(), indices.long(), torch.ones(indices.numel()), size=(n, n)) @ feat
    for hybrid in ("0", "1"):
        assert torch.equal(outs["0"][hybrid], outs["1"][hybrid])
        assert torch.equal(outs["1"][hybrid], want)          # integer operands: exact
