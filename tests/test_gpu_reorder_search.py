"""The Cuthill-McKee search in HIP (voltrix/include/voltrix/reorder_kernels.hpp through the C-ABI) against the plain-loop
restatement of its specification (oracle/oracle_np.py::cm_order) and against the torch host form, element for element.
No reference counterpart: the reference reads externally reordered graphs (bench/graph_gen.py:42-45)."""
import numpy as np
import pytest
import torch

import synth_graphs
from oracle import oracle_np
from voltrix import reorder

pytestmark = pytest.mark.gpu


def _random_csr(rng, n, m, density, duplicates=True):
    a = rng.random((n, m)) < density
    rows = [np.nonzero(r)[0] for r in a]
    if duplicates:      # unsorted rows with a repeated entry now and then: the search must not care
        rows = [rng.permutation(np.concatenate([r, r[:1]])) if len(r) and rng.random() < 0.3 else r for r in rows]
    ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ix = (np.concatenate(rows) if ip[-1] else np.zeros(0)).astype(np.int32)
    return torch.from_numpy(ip), torch.from_numpy(ix)


@pytest.mark.parametrize("seed", range(6))
def test_small_graphs_match_the_specification(cuda_device, seed):
    """Square and rectangular patterns, several components, isolated rows, duplicates, component budgets 1 / 3 / 64."""
    rng = np.random.default_rng(seed)
    for _ in range(12):
        n = int(rng.integers(1, 200))
        m = int(rng.choice([n, n, n + 9, max(1, n - 11)]))
        ip, ix = _random_csr(rng, n, m, float(rng.choice([0.005, 0.02, 0.1])))
        budget = int(rng.choice([1, 3, 64]))
        want = oracle_np.cm_order(ip.numpy(), ix.numpy(), n, m, max_components=budget)
        got = reorder.bfs_permutation(ip.cuda(), ix.cuda(), n, m, max_components=budget).cpu().numpy()
        assert np.array_equal(want, got), (n, m, budget)


def test_wide_levels_take_the_whole_chip_kernels(cuda_device):
    """A sparse random graph of 30 k nodes: levels of many thousand nodes (frontier above the single-workgroup limit of
    2048, levels above the LDS-sort limit of 1024 -> radix sort), a giant component + small ones + isolated rows."""
    rng = np.random.default_rng(7)
    n, deg = 30000, 3
    cols = rng.integers(0, n, size=(n, deg))
    cols[rng.random(n) < 0.05] = -1                    # some rows without entries
    rows = [np.unique(c[c >= 0]) for c in cols]
    ip = torch.from_numpy(np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32))
    ix = torch.from_numpy(np.concatenate(rows).astype(np.int32))
    info = {}
    got = reorder.bfs_permutation(ip.cuda(), ix.cuda(), n, info=info).cpu().numpy()
    want = oracle_np.cm_order(ip.numpy(), ix.numpy(), n)
    assert np.array_equal(want, got)
    host = reorder.bfs_permutation(ip, ix, n).numpy()
    assert np.array_equal(host, got)
    assert info["components"] >= 1


def test_deep_band_graph_is_walked_by_single_launches(cuda_device):
    """A shuffled band graph has hundreds of narrow levels: the single-workgroup kernels walk them (one host read per
    search, not per level), the order equals the host form's, and the TC-block count returns to the natural order's."""
    torch.manual_seed(3)
    n, band = 20000, 24
    rows = torch.arange(n).repeat_interleave(6)
    cols = (rows + torch.randint(-band, band + 1, (rows.numel(),))).clamp(0, n - 1)
    key = torch.unique(rows * n + cols)
    rows, cols = key // n, key % n
    label = torch.randperm(n)
    r2, c2 = label[rows], label[cols]
    order = torch.argsort(r2 * n + c2)
    r2, c2 = r2[order], c2[order]
    ip = torch.zeros(n + 1, dtype=torch.int64)
    ip[1:] = torch.cumsum(torch.bincount(r2, minlength=n), 0)
    ip, ix = ip.to(torch.int32), c2.to(torch.int32)
    info = {}
    got = reorder.bfs_permutation(ip.cuda(), ix.cuda(), n, info=info)
    host = reorder.bfs_permutation(ip, ix, n)
    assert torch.equal(got.cpu(), host)
    assert info["level_reads"] <= 4 * info["components"], info      # hundreds of levels, a handful of host reads
    pip, pix = reorder.permute_rows_csr(ip.cuda(), ix.cuda(), n, got)
    import voltrix

    t_shuffled = int(voltrix.csr_preprocess(ip, ix, n)[0][-1])
    t_ordered = int(voltrix.csr_preprocess(pip.cpu(), pix.cpu(), n)[0][-1])
    assert t_ordered < 0.6 * t_shuffled


def test_reddit_like_sample_matches_the_host_form(cuda_device):
    """Two or three levels holding the whole graph (half of the edges are uniformly random): the whole-chip path at a
    realistic degree (scale 0.05: 11.6 k rows, 2.9 M entries) against the torch host form."""
    ip, ix, _ = synth_graphs.generate("reddit_like", scale=0.05)
    n = ip.numel() - 1
    got = reorder.bfs_permutation(ip.cuda(), ix.cuda(), n).cpu()
    host = reorder.bfs_permutation(ip, ix, n)
    assert torch.equal(got, host)


def test_transpose_keeps_duplicates_and_handles_empty_inputs(cuda_device):
    """capi.csr_transpose (expanded row ids, stable radix sort by column, row pointers by binary search) against the
    (column, row) sort of the entries: unsorted rows, duplicates, empty columns at both ends, no entries at all."""
    from voltrix import capi

    rng = np.random.default_rng(2)
    for n, m, density in ((1, 1, 1.0), (70, 333, 0.02), (513, 64, 0.3), (40, 40, 0.0)):
        ip, ix = _random_csr(rng, n, m, density)
        if ix.numel():
            ix = ix.clamp(min=3, max=m - 2) if m > 8 else ix         # empty columns at both ends
        t_ip, t_ix = capi.csr_transpose(ip.cuda(), ix.cuda(), n, m)
        rows = np.repeat(np.arange(n), np.diff(ip.numpy()))
        order = np.lexsort((rows, ix.numpy()))
        assert np.array_equal(t_ix.cpu().numpy(), rows[order])
        want_ptr = np.concatenate([[0], np.cumsum(np.bincount(ix.numpy(), minlength=m))])
        assert np.array_equal(t_ip.cpu().numpy(), want_ptr)
