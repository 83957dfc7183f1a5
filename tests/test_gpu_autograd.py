"""GPU: the SpMM's backward (A^T @ dC through a transposed handle, voltrix/autograd.py) against torch autograd on the
reference's own oracle expression ``csr(ones) @ feat``."""
import numpy as np
import pytest
import torch

import voltrix
from conftest import load_csr_fixture
from test_hybrid_plan import _random_csr
from voltrix.autograd import SpMM, csr_transpose_device

pytestmark = pytest.mark.gpu


def test_transpose_is_the_transpose(cuda_device):
    indptr, indices = _random_csr(300, 20, seed=3, ncols=457)
    t_indptr, t_indices = csr_transpose_device(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), 300, 457)
    import scipy.sparse as sp

    a = sp.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(300, 457))
    at = a.T.tocsr()
    at.sort_indices()
    assert np.array_equal(t_indptr.cpu().numpy(), at.indptr) and np.array_equal(t_indices.cpu().numpy(), at.indices)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_gradient_matches_torch_autograd(cuda_device, dtype, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    op = SpMM(indptr, indices, n, hash_tag=f"autograd_{dtype}")
    torch.manual_seed(0)
    feat32 = torch.randn(n, 48)
    weight = torch.randn(n, 48)
    feat = feat32.to(dtype).cuda().requires_grad_(True)
    out = op(feat)
    (out * weight.cuda()).sum().backward()
    assert feat.grad is not None and feat.grad.dtype == dtype and feat.grad.shape == (n, 48)
    ref_in = feat32.to(dtype).float().requires_grad_(True)
    a = torch.sparse_csr_tensor(indptr, indices, torch.ones(indices.numel()), size=(n, n))
    ref_out = torch.sparse.mm(a.to_sparse_coo(), ref_in)     # coo: the CPU csr @ dense has no autograd formula
    (ref_out * weight).sum().backward()
    assert float((out.detach().cpu() - ref_out.detach()).norm() / ref_out.norm()) < 1e-3
    # dC is rounded to fp16 (scaled) in the backward SpMM exactly as B is in the forward one
    rel = float((feat.grad.float().cpu() - ref_in.grad).norm() / ref_in.grad.norm())
    assert rel < (2e-3 if dtype == torch.float16 else 1e-3), rel


def test_rectangular_operator_and_exact_integers(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices = _random_csr(260, 30, seed=9, ncols=500)
    op = SpMM(torch.from_numpy(indptr), torch.from_numpy(indices), 260, num_cols=500, hash_tag="autograd_rect")
    feat = torch.randint(-3, 4, (500, 32)).float().cuda().requires_grad_(True)
    out = op(feat)
    assert out.shape == (260, 32)
    out.sum().backward()
    col_deg = np.bincount(indices, minlength=500).astype(np.float32)
    assert torch.equal(feat.grad.cpu(), torch.from_numpy(col_deg)[:, None].expand(500, 32))   # A^T @ 1 = in-degree, exact
