"""GPU: the library's unit-table builder (voltrix/unit_table.hpp, the entry points a C host binds) against the torch-tensor
restatement voltrix.schedule.unit_table_torch -- every array, element by element -- and the table's invariants."""
import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from voltrix import capi
from voltrix.schedule import unit_table, unit_table_torch

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a.num_units == b.num_units and a.num_cuts == b.num_cuts and a.num_slots == b.num_slots
    assert a.max_units_per_xcd == b.max_units_per_xcd and a.max_stages == b.max_stages
    assert torch.equal(a.unit_ptr.cpu(), b.unit_ptr.cpu())
    assert torch.equal(a.units.cpu(), b.units.cpu())
    assert torch.equal(a.cuts.cpu(), b.cuts.cpu())


@pytest.mark.parametrize("name,scale", [("reddit_like", 0.1), ("products_like", 0.05), ("powerlaw_4m", 0.01),
                                        ("reddit_uniform", 0.03)])
@pytest.mark.parametrize("max_stages", [None, 1, 8, 37, 1 << 20])
def test_native_unit_table_equals_the_torch_restatement(cuda_device, name, scale, max_stages):
    indptr, indices, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n = indptr.numel() - 1
    blk_offsets = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[0]
    native = unit_table(blk_offsets, n, max_stages)
    ref = unit_table_torch(blk_offsets, n, max_stages)
    _same(native, ref)
    # invariants: every stage of every window is in exactly one unit; XCD ranges are sorted longest first
    units = native.units.cpu().numpy().astype(np.int64)
    nblk = np.diff(blk_offsets.cpu().numpy().astype(np.int64))
    nst = (nblk + 3) // 4
    w, j, k = units[:, 0], units[:, 1], units[:, 2]
    length = (nst[w] - j + k - 1) // k
    assert (np.bincount(w, weights=length, minlength=nst.size) == nst).all()
    assert (length <= native.max_stages).all() or max_stages is None and native.max_stages >= 8
    ptr = native.unit_ptr.cpu().numpy()
    for x in range(8):
        seg = length[ptr[x]:ptr[x + 1]]
        assert (np.diff(seg) <= 0).all()


def test_native_unit_table_edge_cases(cuda_device):
    dev = torch.device("cuda")
    # no windows
    t = unit_table(torch.zeros(1, dtype=torch.int32, device=dev), 0)
    assert t.num_units == 0 and t.units.shape == (0, 4) and int(t.unit_ptr.abs().sum()) == 0
    # fewer windows than XCDs, empty windows (one all-zero TC block each), one giant window
    for counts in ([1], [1, 1, 1], [5, 1, 4000, 1, 9], list(range(1, 40))):
        blk = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=dev)
        n = 16 * len(counts) - 3
        for max_stages in (None, 2, 100):
            _same(unit_table(blk, n, max_stages), unit_table_torch(blk, n, max_stages))


def test_workspace_sizes_and_bad_arguments(cuda_device):
    lib = capi.lib()
    import ctypes

    assert lib.voltrix_unit_table_workspace_bytes(ctypes.c_int(232965)) > 6 * 4 * 14561
    assert lib.voltrix_unit_table_fill_workspace_bytes(ctypes.c_int64(16565)) > 5 * 4 * 16565
    rc = ctypes.c_int(-1)
    z = ctypes.c_void_p(0)
    lib.voltrix_launch_unit_table_count(z, ctypes.c_int(-1), ctypes.c_int(0), z, z, z, ctypes.byref(rc))
    assert rc.value == 1
    lib.voltrix_launch_unit_table_count(z, ctypes.c_int(64), ctypes.c_int(0), z, z, z, ctypes.byref(rc))
    assert rc.value == 1   # no header


@pytest.mark.parametrize("num_panels", [1, 5, 8, 9, 455, 4100])
def test_native_panel_order_equals_the_torch_restatement(cuda_device, num_panels):
    from voltrix.hybrid import longest_first_order

    g = torch.Generator().manual_seed(num_panels)
    nks = torch.randint(0, 7 if num_panels > 100 else 1000, (num_panels,), generator=g)   # many ties in the large cases
    panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
    panel_ptr[1:] = nks.cumsum(0)
    native = longest_first_order(panel_ptr.cuda())
    ref = longest_first_order(panel_ptr)          # CPU tensor -> torch argsort
    assert native.is_cuda and torch.equal(native.cpu(), ref)
    assert torch.equal(torch.sort(native.cpu().long()).values, torch.arange(num_panels))
