"""GPU: wide operands as one launch per column slab (spmm_kernels.hpp::slab_launches, round 3) against one grid over all
slabs -- same bits for the window format and for the two-level pair, and both equal to the dense product."""
import pytest
import torch

import synth_graphs

pytestmark = pytest.mark.gpu


def test_one_launch_per_slab_gives_the_bits_of_the_single_grid(cuda_device, monkeypatch):
    """The slab policy is an ARGUMENT of every launch (jit_kernels/spmm.py::SLAB_POLICY -> launch_spmm_tc16 / the panel
    entry points): forced to one grid and to one launch per slab group, window format and two-level pair."""
    import voltrix
    from voltrix.jit_kernels import spmm as wrapper

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.04)
    n = indptr.numel() - 1
    torch.manual_seed(0)
    feat = torch.randint(-4, 5, (n, 320)).half()       # 2.5 slabs of 128 columns; integers: every summation order is exact
    want = torch.sparse_csr_tensor(indptr.long(), indices.long(), torch.ones(indices.numel()), size=(n, n)) @ feat.float()
    outs = {}
    for hybrid in ("0", "1"):
        monkeypatch.setenv("VOLTRIX_HYBRID", hybrid)
        monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
        handle = voltrix.csr_preprocess(indptr, indices, n)
        handle[1].hash_tag = f"slab_launches_{hybrid}"
        assert (voltrix.two_level_of(handle[1]) is not None) == (hybrid == "1")
        for policy in (0, 1, -1):
            monkeypatch.setattr(wrapper, "SLAB_POLICY", policy)
            outs[hybrid, policy] = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda()).cpu()
        assert wrapper.slab_launches(320, 128, 2, n, 0) == 1 and wrapper.slab_launches(320, 128, 2, n, 1) == 3
    for hybrid in ("0", "1"):
        assert torch.equal(outs[hybrid, 0], outs[hybrid, 1]) and torch.equal(outs[hybrid, -1], outs[hybrid, 1])
        assert torch.equal(outs[hybrid, 1], want)          # integer operands: exact


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
