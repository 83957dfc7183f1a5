"""CPU: the stream kernel's schedule tables (voltrix/schedule.py::stream_tables) -- every stage of every window is scheduled
exactly once, runs tile the units in window order with at most 64 units each, XCD ranges tile the runs, cut windows get
consecutive partial-tile slots, and the per-window count of edge-carrying columns in the last TC block follows the format."""
import numpy as np
import pytest
import torch

from conftest import load_csr_fixture
from oracle import oracle_c
from voltrix.schedule import FEW_WINDOWS, default_max_stages, last_block_columns, stream_tables, unit_table_torch


def _handle(g):
    n = int(g["num_nodes"])
    p1, packed, hind = oracle_c.csr_preprocess(np.asarray(g["indptr"], np.int32), np.asarray(g["indices"], np.int32), n)
    return torch.from_numpy(p1), torch.from_numpy(packed.view(np.int32)).view(torch.int32), torch.from_numpy(hind), n


@pytest.mark.parametrize("run_cost,cut", [(None, None), (6, 6), (12, 3), (48, 200), (2, 1)])
def test_stream_tables_cover_every_stage_once(csr_fixture, run_cost, cut):
    off, packed, hind, n = _handle(csr_fixture)
    t = stream_tables(off, packed, hind, n, run_cost=run_cost, cut_stages=cut)
    u, r = t.units.numpy(), t.runs.numpy()
    num_windows = (n + 15) // 16
    seen = {}
    for first, end, step, w, slot, length, ncl, _ in u:
        assert off[w] <= first < end == off[w + 1] and step % 4 == 0 and step >= 4
        blocks = list(range(first, end, step))
        assert len(blocks) == length
        for b in blocks:
            seen[(w, b)] = seen.get((w, b), 0) + 1
    want = {(w, b) for w in range(num_windows) for b in range(int(off[w]), int(off[w + 1]), 4)}
    assert set(seen) == want and all(v == 1 for v in seen.values())
    assert (np.diff(u[:, 3]) >= 0).all()                                       # window order
    assert r[:, 1].sum() == len(u) and (r[:, 0] == np.concatenate([[0], np.cumsum(r[:, 1])[:-1]])).all()
    assert r[:, 1].max() <= 64 and r[:, 1].min() >= 1
    assert (r[:, 2] == [u[a:a + c, 5].sum() for a, c, _, _ in r]).all()
    rp = t.run_ptr.numpy()
    assert rp[0] == 0 and rp[8] == t.num_runs == len(r) and (np.diff(rp) >= 0).all() and np.diff(rp).max() == t.max_runs_per_xcd
    cuts = t.cuts.numpy()
    slots = u[u[:, 4] >= 0]
    assert t.num_slots == len(slots) and (np.sort(slots[:, 4]) == np.arange(len(slots))).all()
    for w, first_slot, k, _ in cuts:
        mine = u[u[:, 3] == w]
        assert len(mine) == k and (mine[:, 4] == first_slot + np.arange(k)).all()
    assert t.num_cuts == len(cuts)


def test_last_block_columns_match_the_bitmaps(csr_fixture):
    off, packed, hind, n = _handle(csr_fixture)
    ncl = last_block_columns(off, packed, hind, n).numpy()
    indptr, indices = np.asarray(csr_fixture["indptr"]), np.asarray(csr_fixture["indices"])
    for w in range((n + 15) // 16):
        cols = np.unique(indices[indptr[16 * w]:indptr[min(n, 16 * w + 16)]])
        want = 0 if len(cols) == 0 else (len(cols) - 1) % 8 + 1
        assert ncl[w] == want, w


def test_few_long_windows_are_cut_to_about_one_unit_per_simd():
    """Handles of fewer than FEW_WINDOWS windows: the default unit length is at most ceil(all stages / FEW_WINDOWS) (floor 8), so
    a few hundred windows of one length become about FEW_WINDOWS units; handles with more windows, or whose windows are short,
    keep 1.5 x the median."""
    def offsets(stages_per_window):
        return torch.tensor(np.concatenate([[0], np.cumsum(4 * np.asarray(stages_per_window))]), dtype=torch.int32)

    few = offsets([115] * 267)                       # ddi-like: nothing is longer than 1.5 x the median
    n = 16 * 267
    assert default_max_stages(few, n) == -(-115 * 267 // FEW_WINDOWS) == 30
    t = unit_table_torch(few, n)
    assert t.max_stages == 30 and t.num_units == 267 * 4 and t.num_cuts == 267 and t.num_slots == t.num_units
    assert default_max_stages(offsets([2] * 170), 16 * 170) == 8                     # short windows: the floor
    assert default_max_stages(offsets([115] * FEW_WINDOWS), 16 * FEW_WINDOWS) == 172   # enough windows: 1.5 x the median
    assert default_max_stages(offsets([40] * 900 + [4000] * 100), 16 * 1000) == 60     # the cap never raises the bound
    long_bins = offsets([70000] * 3)                 # stages past the device histogram's last bin count as 65536
    assert default_max_stages(long_bins, 48) == -(-3 * 65536 // FEW_WINDOWS)
