"""GPU: the two-level product in ONE launch (spmm_fused_kernel through the C-ABI entry voltrix_launch_spmm_fused_*) against
the oracle (torch.sparse.mm on the CPU, the reference's own oracle call), against the round-2 pair of kernels bit for bit
on integer operands, and run to run."""
import numpy as np
import pytest
import torch

import synth_graphs
from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _random_csr(n, max_deg, seed, ncols=None):
    rng = np.random.default_rng(seed)
    ncols = ncols or n
    rows = [np.unique(rng.integers(0, ncols, rng.integers(0, max_deg + 1))) for _ in range(n)]
    indptr = np.zeros(n + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return torch.from_numpy(indptr), torch.from_numpy(np.concatenate(rows + [np.zeros(0, np.int64)]).astype(np.int32))


@pytest.fixture
def fused_on(monkeypatch):
    monkeypatch.setenv("VOLTRIX_FUSED", "1")
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")


def _two_level(indptr, indices, n, tau, ncols=None):
    import voltrix

    two = voltrix.csr_preprocess_hybrid(indptr, indices, n, num_cols=ncols, tau=tau)
    assert two.fused is not None
    return two


@pytest.mark.parametrize("feat_dim,dtype", [(128, torch.float16), (64, torch.float16), (32, torch.float16), (256, torch.float16),
                                            (128, torch.bfloat16), (200, torch.float16), (128, torch.float32)])
def test_one_launch_matches_the_oracle(fused_on, feat_dim, dtype):
    import voltrix

    indptr, indices = _random_csr(3001, 60, seed=feat_dim)       # tail window (3001 % 16), tail panel, shared + residual
    n = 3001
    two = _two_level(indptr, indices, n, tau=3)
    assert two.plan.num_ksteps > 0 and two.fused.num_records > 0
    torch.manual_seed(1)
    feat = torch.randn(n, feat_dim).to(dtype)
    out = voltrix.spmm_two_level(two, feat.cuda()).cpu()
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    tol = 1e-5 if dtype != torch.float32 else 1e-3          # fp32 features are rounded to scaled fp16
    assert float((out - ref).norm() / ref.norm()) < tol
    assert out.shape == (n, feat_dim)


def test_one_launch_equals_the_pair_bit_for_bit_on_integers(fused_on, monkeypatch):
    import voltrix

    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.05)
    n = indptr.numel() - 1
    two = _two_level(indptr, indices, n, tau=12)                 # tau 12 leaves a real residual at this scale
    assert two.plan.num_ksteps > 0 and two.fused.num_records > 0
    torch.manual_seed(0)
    feat = torch.randint(-3, 4, (n, 128)).half().cuda()
    one = voltrix.spmm_two_level(two, feat)
    again = voltrix.spmm_two_level(two, feat)
    monkeypatch.setenv("VOLTRIX_FUSED", "0")
    pair = voltrix.spmm_two_level(two, feat)
    torch.cuda.synchronize()
    assert torch.equal(one, pair) and torch.equal(one, again)
    exact = torch_ref.spmm(indptr, indices, feat.float().cpu(), n)
    assert torch.equal(one.cpu(), exact)                         # small integers: every partial sum is exact


def test_one_launch_edge_cases(fused_on):
    import voltrix

    # no shared column at all: the fused form is not built; everything shared: no records; empty rows; NaN in an unreferenced row
    indptr, indices = _random_csr(1500, 6, seed=3, ncols=1500)
    two = voltrix.csr_preprocess_hybrid(indptr, indices, 1500, tau=60000)
    assert two.fused is None and two.plan.num_ksteps == 0
    indptr, indices = _random_csr(1024, 300, seed=4)
    two = _two_level(indptr, indices, 1024, tau=1)
    assert two.fused.num_records == 0
    feat = torch.randn(1024, 128).half()
    out = voltrix.spmm_two_level(two, feat.cuda()).cpu()
    ref = torch_ref.spmm(indptr, indices, feat.float(), 1024)
    assert float((out - ref).norm() / ref.norm()) < 1e-5
    indptr, indices = _random_csr(2000, 50, seed=5)
    used = torch.zeros(2000, dtype=torch.bool)
    used[indices.long()] = True
    feat = torch.randn(2000, 128).half()
    feat[~used] = float("nan")
    if used[0]:
        pass
    two = _two_level(indptr, indices, 2000, tau=3)
    out = voltrix.spmm_two_level(two, feat.cuda()).cpu()
    assert torch.isfinite(out).all()                             # padded slots never gather an unreferenced row
    ref = torch_ref.spmm(indptr, indices, torch.nan_to_num(feat.float()), 2000)
    assert float((out - ref).norm() / ref.norm()) < 1e-5


def test_operator_uses_the_one_launch_form_when_asked(fused_on, monkeypatch):
    import voltrix

    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n = indptr.numel() - 1
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = "test_fused_operator"
    two = voltrix.two_level_of(handle[1])
    assert two is not None and two.fused is not None
    feat = torch.randn(n, 128).half()
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda()).cpu()
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert float((out - ref).norm() / ref.norm()) < 1e-5


def test_hip_record_builder_matches_the_definition(fused_on):
    """fused_plan.hpp (through voltrix_launch_fused_records_count / _fill) == the torch-op builder == the plain-loop definition."""
    import voltrix
    from oracle import oracle_np
    from voltrix import hybrid

    cases = [(_random_csr(3001, 60, seed=11), 3001, 3), (_random_csr(700, 9, seed=12), 700, 2)]
    ip, ix, _ = synth_graphs.generate("reddit_like", scale=0.05)
    cases.append(((ip, ix), ip.numel() - 1, 12))
    for (indptr, indices), n, tau in cases:
        two = voltrix.csr_preprocess_hybrid(indptr, indices, n, tau=tau)
        if two.fused is None:      # an empty plan: build the records directly (the builder does not need a plan)
            two.fused = hybrid.build_fused_records(two.blk_offsets, two.hspa_packed, two.hind, n)
        ref = hybrid.build_fused_records_torch(two.blk_offsets, two.hspa_packed, two.hind, n)
        assert two.fused.num_records == ref.num_records
        assert torch.equal(two.fused.wave_ptr, ref.wave_ptr)
        assert torch.equal(two.fused.records.view(torch.int32), ref.records.view(torch.int32))
        if n <= 3001:
            wp, rec = oracle_np.fused_records(two.blk_offsets.cpu().numpy(), two.hspa_packed.cpu().view(torch.int32).numpy().view(np.uint32),
                                              two.hind.cpu().numpy(), n)
            assert np.array_equal(two.fused.wave_ptr.cpu().numpy(), wp)
            assert np.array_equal(two.fused.records.cpu().view(torch.int32).numpy().view(np.uint32), rec)
