"""GPU: weighted SpMM (value plane beside the bitmaps, WEIGHTED tiles of the window kernel) against the reference's
oracle expression with real values: ``torch.sparse_csr_tensor(indptr, indices, values) @ feat`` on the CPU.

Tolerance: A values and B are both rounded to the 16-bit operand type before the MFMA (fp32 accumulate), so against the
fp32 oracle the element-wise bound is (2^-10 + deg 2^-23) (|A| |B|) (two roundings of 2^-11 each, first order) plus the
accumulation term; against the oracle evaluated on the SAME rounded operands only deg 2^-23 (|A| |B|) remains."""
import numpy as np
import pytest
import torch

import voltrix
from conftest import load_csr_fixture
from test_hybrid_plan import _random_csr
from voltrix.weighted import value_plane

pytestmark = pytest.mark.gpu


def _oracle(indptr, indices, values, feat, n, ncols):
    a = torch.sparse_csr_tensor(torch.as_tensor(indptr), torch.as_tensor(indices), values.double(), size=(n, ncols))
    return (a @ feat.double())


def test_value_plane_follows_the_format_definition(cuda_device, csr_fixture):
    g = csr_fixture
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    values = torch.arange(1, len(indices) + 1, dtype=torch.float32)
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    plane = h.values32.cpu().numpy()
    hind = h.hind.cpu().numpy().reshape(-1, 8)
    p1 = h.blk_offsets.cpu().numpy()
    want = np.zeros_like(plane)
    for r in range(n):
        w = r // 16
        for e in range(g["indptr"][r], g["indptr"][r + 1]):
            c = g["indices"][e]
            blocks = hind[p1[w]:p1[w + 1]]
            cols = sorted(set(g["indices"][g["indptr"][16 * w]:g["indptr"][min(n, 16 * w + 16)]].tolist()))
            q = cols.index(c)
            assert blocks[q // 8, q % 8] == c
            want[p1[w] + q // 8, r % 16, q % 8] += values[e].item()      # duplicates add
    assert np.array_equal(plane, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("width", [32, 128, 200])
def test_weighted_spmm_matches_the_oracle(cuda_device, dtype, width, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    torch.manual_seed(3)
    values = torch.randn(len(indices))
    feat32 = torch.randn(n, width)
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    out = voltrix.spmm_weighted(h, feat32.to(dtype).cuda(), hash_tag=f"weighted_{width}")
    assert out.shape == (n, width) and out.dtype == torch.float32 and not torch.isnan(out).any()
    ref = _oracle(g["indptr"], g["indices"], values, feat32.to(dtype) if dtype != torch.float32 else feat32, n, n)
    absref = _oracle(g["indptr"], g["indices"], values.abs(), feat32.abs(), n, n)
    deg = torch.from_numpy(np.diff(g["indptr"]).astype(np.float64))
    u = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    bound = (u + (deg[:, None] + 1) * 2.0 ** -23) * absref + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    assert float((out.cpu().double() - ref).norm() / ref.norm()) < (2e-2 if dtype == torch.bfloat16 else 2e-3)
    # same rounded operands on both sides: accumulation order only
    vdt = torch.float16 if dtype == torch.float32 else dtype
    if dtype != torch.float32:
        ref_same = _oracle(g["indptr"], g["indices"], values.to(vdt).float(), feat32.to(dtype).float(), n, n)
        assert ((out.cpu().double() - ref_same).abs() <= (deg[:, None] + 1) * 2.0 ** -23 * absref + 1e-6).all()
    assert torch.equal(out, voltrix.spmm_weighted(h, feat32.to(dtype).cuda()))


def test_weighted_with_unit_values_is_the_binary_product_and_duplicates_add(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    indptr, indices = _random_csr(900, 60, seed=4)
    ip, ix = torch.from_numpy(indptr), torch.from_numpy(indices)
    feat = torch.randint(-3, 4, (900, 64)).half().cuda()
    h = voltrix.csr_preprocess_weighted(ip, ix, torch.ones(len(indices)), 900)
    plain = voltrix.csr_preprocess(ip, ix, 900)
    plain[1].hash_tag = "weighted_vs_binary"
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((h.blk_offsets, h.hspa_packed, h.hind), plain))
    assert torch.equal(voltrix.spmm_weighted(h, feat, hash_tag="weighted_unit"),
                       voltrix.spmm(*plain, num_nodes=900, num_edges=len(indices), feat=feat))
    # duplicate entries: values add (torch.sparse.mm semantics), the pattern counts them once
    dup_indices = torch.cat([ix[:3], ix])          # row 0 gets its first three columns twice (if it has three)
    if indptr[1] >= 3:
        dup_indptr = ip.clone()
        dup_indptr[1:] += 3
        vals = torch.ones(len(dup_indices))
        hd = voltrix.csr_preprocess_weighted(dup_indptr, dup_indices, vals, 900)
        out = voltrix.spmm_weighted(hd, torch.ones(900, 8).half().cuda(), hash_tag="weighted_dup")
        assert float(out[0, 0]) == float(indptr[1]) + 3.0


def test_value_plane_in_chunks_and_without_an_fp32_master(cuda_device, monkeypatch):
    """Round 5: the plane is built window chunk by window chunk (bounded temporaries) -- the same bits as in one piece -- and a
    handle too large for an fp32 master keeps only the 16-bit plane it was built with."""
    from voltrix import weighted

    indptr_np, indices_np = _random_csr(2500, 40, seed=11)
    indptr, indices = torch.from_numpy(indptr_np), torch.from_numpy(indices_np)
    n = 2500
    torch.manual_seed(5)
    values = torch.randn(len(indices_np))
    whole = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    for chunk in (1, 1000, 7777):
        monkeypatch.setattr(weighted, "CHUNK_EDGES", chunk)
        again = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
        assert torch.equal(again.values32, whole.values32), chunk
    monkeypatch.setattr(weighted, "MASTER_PLANE_MAX_BYTES", 0)
    lean = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    assert lean.values32 is None and list(lean.planes) == [torch.float16]
    assert torch.equal(lean.planes[torch.float16], whole.values32.half())
    feat = torch.randn(n, 64, device=cuda_device).half()
    assert torch.equal(voltrix.spmm_weighted(lean, feat, hash_tag="lean"), voltrix.spmm_weighted(whole, feat, hash_tag="lean"))
    with pytest.raises(AssertionError, match="without an fp32 master"):
        voltrix.spmm_weighted(lean, feat.bfloat16())
    lean_bf = voltrix.csr_preprocess_weighted(indptr, indices, values, n, plane_dtype=torch.bfloat16)
    assert list(lean_bf.planes) == [torch.bfloat16] and torch.equal(lean_bf.planes[torch.bfloat16], whole.values32.bfloat16())
