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


@pytest.mark.parametrize("dtype,mode,width", [(torch.float32, "exact", 200), (torch.bfloat16, "fp16", 320), (torch.float32, "fp16", 264)])
def test_wide_operands_of_every_dtype_against_the_dense_product(cuda_device, dtype, mode, width, monkeypatch):
    """Several column slabs (one launch each on a graph this small) for the exact-fp32 tiles (64 columns x 4 bytes), the
    bfloat16 operand and the scaled-fp16 path of fp32 features, widths that are not multiples of the slab."""
    import voltrix

    monkeypatch.setenv("VOLTRIX_FP32_MODE", mode)
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n = indptr.numel() - 1
    torch.manual_seed(1)
    feat = torch.randint(-3, 4, (n, width)).to(dtype)          # small integers: exact in every operand type
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = f"wide_{dtype}_{mode}_{width}"
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda())
    want = torch.sparse_csr_tensor(indptr.long(), indices.long(), torch.ones(indices.numel()), size=(n, n)) @ feat.float()
    assert torch.equal(out.cpu(), want)
