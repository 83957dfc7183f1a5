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


def test_two_layer_gcn_gradients_match_a_dense_reference(cuda_device, monkeypatch):
    """examples/gcn_train.py end to end on a small graph: loss and every parameter gradient of the two-layer GCN (normalised
    adjacency with self loops, separable values on the binary operator, both directions) against the same model on a dense fp64 Â."""
    import importlib.util
    import os

    import synth_graphs

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    spec = importlib.util.spec_from_file_location("gcn_train", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "examples", "gcn_train.py"))
    gcn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gcn)
    indptr, indices, _ = synth_graphs.generate("com_amazon_like", device="cuda", scale=0.004)
    n = indptr.numel() - 1
    indptr, indices, values = gcn.normalised_adjacency(indptr, indices, n)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (indptr[1:] - indptr[:-1]).long())
    assert int((rows == indices.long()).sum()) == n                     # one self loop per node
    op = SpMM(indptr, indices, n, values=values, hash_tag="gcn_example_test")
    assert op.weighted.separable
    dense = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    dense[rows, indices.long()] = values.double()
    torch.manual_seed(1)
    x = torch.randn(n, 24, device="cuda")
    y = torch.randint(0, 8, (n,), device="cuda")
    model = gcn.GCN(op, 24, 32, 8).cuda()
    ref = gcn.GCN(lambda h: dense @ h, 24, 32, 8, dtype=torch.float64).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    loss = torch.nn.functional.cross_entropy(model(x), y)
    loss.backward()
    ref_loss = torch.nn.functional.cross_entropy(ref(x.double()), y)
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) <= 2e-3 * abs(float(ref_loss))
    for (name, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        err = float((p.grad.double() - q.grad).norm() / q.grad.norm())
        assert err <= 1e-2, (name, err)              # four fp16 operand roundings along each gradient path
