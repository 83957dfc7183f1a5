"""GPU: the CSR row-gather kernel (spmm_csr_kernels.hpp, round 6) and the operator's measured choice between it and the block-format
path.  Oracle: the reference's own expression ``torch.sparse_csr_tensor(indptr, indices, ones) @ feat`` on the CPU
(tests/test_spmm.py:24-29 of the reference).  Binary A: products are exact, the sum is fp32 in entry order:
|C - ref| <= deg 2^-23 (A |B|) against the oracle on the SAME operand; integers: exact."""
import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from conftest import load_csr_fixture
from oracle import torch_ref
from voltrix import capi, sidecar

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("width", [8, 32, 72, 128, 200, 512, 1024])
def test_csr_kernel_matches_the_oracle(cuda_device, csr_fixture, dtype, width):
    g = csr_fixture
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]).cuda(), torch.from_numpy(g["indices"]).cuda()
    torch.manual_seed(width)
    feat = torch.randn(n, width).to(dtype)
    out = torch.full((n, width), float("nan"), device=cuda_device)
    capi.launch_spmm_csr_rows(indptr, indices, n, feat.cuda(), out, torch.cuda.current_stream().cuda_stream)
    ref = torch_ref.spmm(g["indptr"], g["indices"], feat.float(), n).double()
    absref = torch_ref.spmm(g["indptr"], g["indices"], feat.float().abs(), n).double()
    deg = torch.from_numpy(np.diff(g["indptr"]).astype(np.float64))[:, None]
    assert not torch.isnan(out).any()                                   # every row written, empty rows as zeros
    assert ((out.cpu().double() - ref).abs() <= (deg + 1) * 2.0 ** -23 * absref + 1e-30).all()
    ints = torch.randint(-3, 4, (n, width)).to(dtype)
    capi.launch_spmm_csr_rows(indptr, indices, n, ints.cuda(), out, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out.cpu(), torch_ref.spmm(g["indptr"], g["indices"], ints.float(), n))


@pytest.mark.parametrize("graph,scale", [("dd_like", 0.2), ("com_amazon_like", 0.2), ("yeast_like", 0.05)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_operator_keeps_the_csr_for_short_window_handles_and_both_paths_agree(cuda_device, graph, scale, dtype, monkeypatch):
    """csr_preprocess attaches the device CSR to handles of short windows; VOLTRIX_CSR_PATH=1 / 0 force either path, auto times
    them once per (width, dtype) and remembers; all three give the oracle's product (integers: bit for bit)."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "default")         # "none" / "stream" pin the block-format kernel: no measured choice then
    monkeypatch.delenv("VOLTRIX_FP32_MODE", raising=False)
    indptr, indices, _ = synth_graphs.generate(graph, scale=scale)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = None
    csr = sidecar.lookup_csr(handle[1])
    assert csr is not None and csr.num_rows == n and csr.indices.numel() == e
    feat = torch.randint(-3, 4, (n, 96)).to(dtype).cuda()
    ref = torch_ref.spmm(indptr.numpy(), indices.numpy(), feat.float().cpu(), n)
    got = {}
    for mode in ("1", "0", "auto"):
        monkeypatch.setenv("VOLTRIX_CSR_PATH", mode)
        got[mode] = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
        assert torch.equal(got[mode].cpu(), ref), mode
    assert csr.choice.get((96, str(dtype))) in ("csr", "block")            # auto decided, once
    before = dict(csr.choice)
    voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    assert csr.choice == before
    # pinned kernels / numerics stay pinned: no measured choice, the block-format path
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    other = torch.randint(-3, 4, (n, 40)).to(dtype).cuda()
    assert torch.equal(voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=other).cpu(),
                       torch_ref.spmm(indptr.numpy(), indices.numpy(), other.float().cpu(), n))
    assert (40, str(dtype)) not in csr.choice


def test_no_csr_side_car_for_duplicates_long_windows_or_when_switched_off(cuda_device, monkeypatch):
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    # duplicate (row, col) entries: the bitmaps count them once, the CSR kernel would count them twice -> block format only
    if int(indptr[1]) >= 2:
        dup_indices = torch.cat([indices[:2], indices])
        dup_indptr = indptr.clone()
        dup_indptr[1:] += 2
        h = voltrix.csr_preprocess(dup_indptr, dup_indices, n)
        assert sidecar.lookup_csr(h[1]) is None
    # a dense graph (long windows): the block format shares columns there
    d_indptr, d_indices, _ = synth_graphs.generate("ddi_like", scale=1.0)
    h = voltrix.csr_preprocess(d_indptr, d_indices, d_indptr.numel() - 1)
    assert sidecar.lookup_csr(h[1]) is None
    monkeypatch.setenv("VOLTRIX_CSR_PATH", "0")
    y_indptr, y_indices, _ = synth_graphs.generate("yeast_like", scale=0.01)
    h = voltrix.csr_preprocess(y_indptr, y_indices, y_indptr.numel() - 1)
    assert sidecar.lookup_csr(h[1]) is None
