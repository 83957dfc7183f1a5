"""CPU: the two-level format's plan in its torch-op form (hybrid.build_panel_plan_torch; the HIP builder is checked the
same way on the GPU, tests/test_gpu_hybrid.py) against the plain-loop oracle
(oracle_np.panel_plan), bit for bit, and the plan's consumer-side interpretation against the input CSR."""
import numpy as np
import pytest
import torch

import synth_graphs
from conftest import load_csr_fixture
from oracle import oracle_np
from voltrix import hybrid


def _random_csr(n, max_deg, seed, ncols=None):
    rng = np.random.default_rng(seed)
    ncols = ncols or n
    rows = [np.unique(rng.integers(0, ncols, rng.integers(0, max_deg + 1))) for _ in range(n)]
    indptr = np.zeros(n + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return indptr, np.concatenate(rows + [np.zeros(0, np.int64)]).astype(np.int32)


def _check(indptr, indices, n, waves, rb, tau, ncols=None):
    ri, rx, plan = hybrid.build_panel_plan_torch(torch.from_numpy(indptr), torch.from_numpy(indices), n, ncols, waves, rb, tau)
    o_ri, o_rx, o_ptr, o_cols, o_bits = oracle_np.panel_plan(indptr, indices, n, waves, rb, tau)
    assert np.array_equal(ri.numpy(), o_ri) and np.array_equal(rx.numpy(), o_rx)
    assert np.array_equal(plan.panel_ptr.numpy(), o_ptr)
    assert np.array_equal(plan.panel_cols.numpy(), o_cols)
    assert np.array_equal(plan.panel_bits.view(torch.int32).numpy().view(np.uint32), o_bits)
    assert plan.num_ksteps == int(o_ptr[-1]) and plan.num_resid_edges == len(o_rx)
    # consumer-side interpretation + residual = the input, each (row, col) once
    shared = oracle_np.panel_to_edges(o_ptr, o_cols, o_bits, n, waves, rb)
    resid = [(r, int(c)) for r in range(n) for c in o_rx[o_ri[r]:o_ri[r + 1]]]
    full = sorted({(r, int(c)) for r in range(n) for c in indices[indptr[r]:indptr[r + 1]]})
    assert sorted(shared + resid) == full
    assert plan.num_shared_edges == len(shared)
    return plan


@pytest.mark.parametrize("waves,rb", [(4, 2), (8, 4), (4, 4), (8, 2)])
@pytest.mark.parametrize("tau", [1, 2, 4])
def test_plan_matches_oracle_random(waves, rb, tau):
    indptr, indices = _random_csr(700, 60, seed=waves * 10 + rb + tau)
    _check(indptr, indices, 700, waves, rb, tau)


def test_plan_on_fixtures(csr_fixture):
    g = csr_fixture
    _check(g["indptr"], g["indices"], int(g["num_nodes"]), 4, 2, 2)


def test_plan_edge_cases():
    # empty matrix, empty panels in the middle, a tail panel, a threshold nothing reaches, duplicates, non-square
    _check(np.zeros(41, np.int32), np.zeros(0, np.int32), 40, 4, 2, 2)
    indptr, indices = _random_csr(300, 30, seed=5)
    indptr2 = indptr.copy()
    indptr2[129:257] = indptr2[128]          # rows 128..255 (panel 1 of 128-row panels) lose their edges ...
    keep = np.r_[0:indptr[128], indptr[256]:indptr[-1]]
    indptr2[257:] -= indptr[256] - indptr[128]
    _check(indptr2, indices[keep], 300, 4, 2, 2)
    plan = _check(indptr, indices, 300, 4, 2, 10 ** 6)
    assert plan.num_ksteps == 0 and plan.num_shared_edges == 0
    dup_indices = np.repeat(indices, 2)      # every entry twice: counts once
    _check((indptr * 2).astype(np.int32), dup_indices, 300, 4, 2, 2)
    indptr3, indices3 = _random_csr(200, 40, seed=9, ncols=1000)
    _check(indptr3, indices3, 200, 4, 2, 2, ncols=1000)


def test_plan_scaled_reddit_like_shares_the_band():
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.01)
    n = indptr.numel() - 1
    _, _, plan = hybrid.build_panel_plan_torch(indptr, indices, n, None, 8, 4, 3)
    assert plan.num_shared_edges + plan.num_resid_edges == indices.numel()
    assert plan.num_shared_edges > 0.3 * indices.numel()


@pytest.mark.parametrize("cap", [1, 7, 64, 10 ** 6])
@pytest.mark.parametrize("num_panels", [1, 9, 200])
def test_panel_part_table(num_panels, cap):
    """Round 4: the panel kernel's launch table in pieces of at most ``cap`` k-steps (hybrid.panel_parts).  Every k-step of
    every panel is in exactly one piece; pieces of a cut panel are contiguous, in order, nearly equal, and take consecutive
    slots; whole panels keep slot -1 (panels without k-steps stay in the table: store mode writes their zero rows); every XCD
    range lists its own panels' pieces longest first."""
    from voltrix import hybrid
    from voltrix.schedule import split_equal_work

    g = torch.Generator().manual_seed(num_panels + cap)
    nks = torch.randint(0, 40, (num_panels,), generator=g) * (torch.arange(num_panels) % 5 == 0).long() * 11 + \
        torch.randint(0, 9, (num_panels,), generator=g)
    panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
    panel_ptr[1:] = nks.cumsum(0)
    for xcd_ptr in (None, split_equal_work(nks)):
        t = hybrid.panel_parts(panel_ptr, cap, xcd_ptr)
        parts, cuts = t.parts.numpy().astype(np.int64), t.cuts.numpy().astype(np.int64)
        assert t.num_parts == len(parts) >= num_panels and t.num_cuts == len(cuts) and t.cap == cap
        assert (parts[:, 2] <= cap).all() and (parts[:, 2] >= 0).all()
        covered = [np.zeros(int(x), np.int64) for x in nks]
        for p, b, n, _ in parts:
            covered[p][b:b + n] += 1
        assert all((c == 1).all() for c in covered)
        slots = 0
        for p, first, pieces, _ in cuts:
            assert first == slots and pieces == -(-int(nks[p]) // cap) >= 2
            mine = parts[parts[:, 0] == p]
            mine = mine[np.argsort(mine[:, 3])]
            assert (mine[:, 3] == first + np.arange(pieces)).all()
            assert (mine[:, 1] == np.concatenate([[0], np.cumsum(mine[:, 2])[:-1]])).all()      # contiguous, in slot order
            assert mine[:, 2].max() - mine[:, 2].min() <= 1
            slots += pieces
        assert slots == t.num_slots == int((parts[:, 3] >= 0).sum())
        whole = parts[parts[:, 3] < 0]
        assert sorted(whole[:, 0]) == sorted(set(range(num_panels)) - set(cuts[:, 0]))
        xp = t.xcd_ptr.numpy()
        assert xp[0] == 0 and xp[8] == t.num_parts and t.max_parts_per_xcd == np.diff(xp).max()
        panel_xcd = (np.searchsorted(xcd_ptr.numpy()[1:8], np.arange(num_panels), side="right") if xcd_ptr is not None
                     else np.arange(num_panels) // max(1, (num_panels + 7) // 8))
        for x in range(8):
            seg = parts[xp[x]:xp[x + 1]]
            assert (panel_xcd[seg[:, 0]] == x).all() and (np.diff(seg[:, 2]) <= 0).all()
