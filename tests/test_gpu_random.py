"""GPU: randomized parity sweep -- many small random CSRs (empty rows, hubs, duplicates, unsorted rows, every N % 16,
odd feature widths) through the operator API; handle bit-exact against the oracle, SpMM within the stated bounds."""
import numpy as np
import pytest
import torch

import voltrix
from oracle import oracle_c, oracle_np, torch_ref

pytestmark = pytest.mark.gpu


def _random_csr(rng, n, ncols_same=True):
    kind = rng.integers(0, 4)
    if kind == 0:      # sparse, many empty rows
        deg = rng.integers(0, 4, n) * (rng.random(n) < 0.3)
    elif kind == 1:    # dense-ish
        deg = rng.integers(0, max(2, n // 3), n)
    elif kind == 2:    # hubs
        deg = rng.integers(0, 6, n)
        deg[rng.choice(n, max(1, n // 20), replace=False)] = rng.integers(n // 2, n + 1)
    else:              # constant degree
        deg = np.full(n, min(n, int(rng.integers(1, 40))))
    deg = np.minimum(deg, n).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    rows = []
    for d in deg:
        c = rng.choice(n, int(d), replace=False)
        if rng.random() < 0.5:
            c = np.sort(c)
        rows.append(c)
    indices = np.concatenate(rows) if rows and indptr[-1] > 0 else np.zeros(0, np.int64)
    return indptr.astype(np.int32), indices.astype(np.int32)


@pytest.mark.parametrize("seed", range(12))
def test_random_graphs_operator_api(cuda_device, seed, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 700))
    num_feats = int(rng.choice([1, 3, 8, 17, 32, 50, 64, 96, 128, 160, 264]))
    indptr, indices = _random_csr(rng, n)
    handle = voltrix.csr_preprocess(torch.from_numpy(indptr), torch.from_numpy(indices), n)
    handle[1].hash_tag = f"rand{seed}"
    op1, opacked, ohind = oracle_c.csr_preprocess(indptr, indices, n)
    assert np.array_equal(handle[0].cpu().numpy(), op1)
    assert np.array_equal(handle[1].cpu().numpy(), opacked)
    assert np.array_equal(handle[2].cpu().numpy(), ohind)

    feat = torch.from_numpy(rng.standard_normal((n, num_feats)).astype(np.float32))
    ref = torch_ref.spmm(indptr, indices, feat, n).numpy().astype(np.float64)
    deg = np.diff(indptr.astype(np.int64)).astype(np.float64)
    aabs = oracle_np.spmm_csr(indptr, indices, np.abs(feat.numpy().astype(np.float64)), n)
    for dtype, mode, u in ((torch.float16, "fp16", 2.0 ** -11), (torch.float32, "fp16", 2.0 ** -11),
                           (torch.float32, "exact", 0.0)):
        monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact" if mode == "exact" else "fp16")
        out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(indices), feat=feat.to(dtype).cuda())
        assert out.shape == (n, num_feats) and out.dtype == torch.float32
        got = out.cpu().numpy().astype(np.float64)
        assert not np.isnan(got).any()
        # operand rounding (relative 2^-11 for normal fp16 values, absolute 2^-25 per edge below the fp16 normal range)
        # + fp32 accumulation in any order
        sub = deg[:, None] * 2.0 ** -25 if u else 0.0
        bound = (u * 1.0001 + deg[:, None] * 2.0 ** -23) * aabs + sub + 1e-30
        assert (np.abs(got - ref) <= bound).all(), (seed, dtype, mode, float(np.abs(got - ref).max()))


def test_duplicate_entries_count_once_like_the_reference_format(cuda_device, monkeypatch):
    """Quirk 5: the bitmap format cannot represent multiplicity; csr(ones) with a duplicated (row, col) gives 2 in
    torch.sparse.mm but 1 here (and in the reference).  Pin the documented behaviour."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr = torch.tensor([0, 3, 4], dtype=torch.int32)
    indices = torch.tensor([1, 1, 0, 1], dtype=torch.int32)
    handle = voltrix.csr_preprocess(indptr, indices, 2)
    handle[1].hash_tag = "dups"
    feat = torch.tensor([[1.0] * 8, [10.0] * 8], dtype=torch.float16)
    out = voltrix.spmm(*handle, num_nodes=2, num_edges=4, feat=feat.cuda()).cpu()
    assert torch.equal(out[:, 0], torch.tensor([11.0, 10.0]))


@pytest.mark.parametrize("seed", range(10))
def test_random_graphs_two_level_format(cuda_device, seed, monkeypatch):
    """The same random graphs (hubs, empty rows, unsorted rows -- canonicalised by the plan builder) through the two-level
    format with random plan geometry and threshold: plan + residual bit-exact against the plain-loop oracle, SpMM within
    the stated bounds."""
    from voltrix import hybrid

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 900))
    num_feats = int(rng.choice([1, 8, 17, 32, 50, 64, 96, 128, 160, 264]))
    waves, rb = [(8, 4), (4, 4), (8, 2), (4, 2)][int(rng.integers(0, 4))]
    tau = int(rng.choice([1, 2, 3, 5]))
    indptr, indices = _random_csr(rng, n)
    handle = voltrix.csr_preprocess_hybrid(torch.from_numpy(indptr), torch.from_numpy(indices), n, waves=waves,
                                           row_blocks=rb, tau=tau)
    handle.hash_tag = f"rand2l{seed}"
    plan = handle.plan
    # canonical form of the input (sorted rows) for the oracle
    s_indices = np.concatenate([np.sort(indices[indptr[r]:indptr[r + 1]]) for r in range(n)] + [np.zeros(0, np.int32)])
    o_ri, o_rx, o_ptr, o_cols, o_bits = oracle_np.panel_plan(indptr, s_indices.astype(np.int32), n, waves, rb, tau)
    assert np.array_equal(plan.panel_ptr.cpu().numpy(), o_ptr) and np.array_equal(plan.panel_cols.cpu().numpy(), o_cols)
    assert np.array_equal(plan.panel_bits.view(torch.int32).cpu().numpy().view(np.uint32), o_bits)
    op1, opacked, ohind = oracle_c.csr_preprocess(o_ri, o_rx, n)     # the residual's handle = the oracle's, bit for bit
    assert np.array_equal(handle.blk_offsets.cpu().numpy(), op1) and np.array_equal(handle.hind.cpu().numpy(), ohind)
    assert np.array_equal(handle.hspa_packed.view(torch.int32).cpu().numpy().view(np.uint32), opacked)

    feat = torch.from_numpy(rng.standard_normal((n, num_feats)).astype(np.float32))
    ref = torch_ref.spmm(indptr, indices, feat, n).numpy().astype(np.float64)
    deg = np.diff(indptr.astype(np.int64)).astype(np.float64)
    aabs = oracle_np.spmm_csr(indptr, indices, np.abs(feat.numpy().astype(np.float64)), n)
    for dtype in (torch.float16, torch.float32):
        out = voltrix.spmm_two_level(handle, feat.to(dtype).cuda(), concurrent=bool(seed % 2))
        assert out.shape == (n, num_feats) and out.dtype == torch.float32
        got = out.cpu().numpy().astype(np.float64)
        assert not np.isnan(got).any()
        bound = (2.0 ** -11 * 1.0001 + (deg[:, None] + 1) * 2.0 ** -23) * aabs + deg[:, None] * 2.0 ** -25 + 1e-30
        assert (np.abs(got - ref) <= bound).all(), (seed, dtype, float(np.abs(got - ref).max()))
