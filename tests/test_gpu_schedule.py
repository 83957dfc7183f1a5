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


def test_few_long_windows_native_table_and_product(cuda_device):
    """Fewer than 1024 windows of one length (ddi-like): the device builder applies the same cap as the restatement, every
    window is cut, and the product through the cut table is exact on integers."""
    dev = torch.device("cuda")
    for counts in ([460] * 267, [7] * 1023, [300] * 1024, [40] * 900 + [16000] * 100, [280000] * 3):
        blk = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=dev)
        n = 16 * len(counts)
        _same(unit_table(blk, n), unit_table_torch(blk, n))
    indptr, indices, _ = synth_graphs.generate("ddi_like", device="cuda")
    n, nnz = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    table = unit_table(handle[0], n)
    assert table.num_cuts == (n + 15) // 16 and 900 <= table.num_units <= 1400
    feat = torch.randint(-3, 4, (n, 96), device="cuda").half()
    ref = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device="cuda"), size=(n, n)) @ feat.float()
    assert torch.equal(voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat), ref)


def test_workspace_sizes_and_bad_arguments(cuda_device):
    lib = capi.lib()
    import ctypes

    assert lib.voltrix_unit_table_workspace_bytes(ctypes.c_int(232965)) > 6 * 4 * 14561
    assert lib.voltrix_unit_table_fill_workspace_bytes(ctypes.c_int64(16565)) > 5 * 4 * 16565
    rc = ctypes.c_int(-1)
    z = ctypes.c_void_p(0)
    lib.voltrix_launch_unit_table_count(z, ctypes.c_int(-1), ctypes.c_int(0), z, z, z, z, ctypes.byref(rc))
    assert rc.value == 1
    lib.voltrix_launch_unit_table_count(z, ctypes.c_int(64), ctypes.c_int(0), z, z, z, z, ctypes.byref(rc))
    assert rc.value == 1   # no header


@pytest.mark.parametrize("name,scale", [("reddit_sbm", 0.1), ("reddit_like", 0.05), ("powerlaw_4m", 0.01)])
@pytest.mark.parametrize("max_stages", [None, 3, 37])
def test_unit_table_over_ranges_of_equal_work(cuda_device, name, scale, max_stages):
    """Round 4: XCD ranges given as data (xcd_ptr, first window of every range) instead of W / 8 windows each.  The native
    builder and the torch restatement agree element by element; every stage is still in exactly one unit; a window's XCD is
    the range that holds it; and the ranges balanced_xcd_windows draws hold about equal stages."""
    from voltrix.schedule import balanced_xcd_windows

    indptr, indices, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n = indptr.numel() - 1
    blk_offsets = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[0]
    w_total = (n + 15) // 16
    for xcd_ptr in (balanced_xcd_windows(blk_offsets, n), balanced_xcd_windows(blk_offsets, n, align=32),
                    torch.tensor([0, 0, 5, 5, w_total // 2, w_total // 2, w_total - 1, w_total, w_total], dtype=torch.int32,
                                 device="cuda")):      # empty ranges, a range of one window
        assert int(xcd_ptr[0]) == 0 and int(xcd_ptr[8]) == w_total and bool((xcd_ptr[1:] >= xcd_ptr[:-1]).all())
        native = unit_table(blk_offsets, n, max_stages, xcd_ptr=xcd_ptr)
        ref = unit_table_torch(blk_offsets, n, max_stages, xcd_ptr=xcd_ptr)
        _same(native, ref)
        units = native.units.cpu().numpy().astype(np.int64)
        nst = (np.diff(blk_offsets.cpu().numpy().astype(np.int64)) + 3) // 4
        w, j, k = units[:, 0], units[:, 1], units[:, 2]
        length = (nst[w] - j + k - 1) // k
        assert (np.bincount(w, weights=length, minlength=nst.size) == nst).all()
        ptr, xp = native.unit_ptr.cpu().numpy(), xcd_ptr.cpu().numpy()
        for x in range(8):
            seg = slice(ptr[x], ptr[x + 1])
            assert ((w[seg] >= xp[x]) & (w[seg] < xp[x + 1])).all() and (np.diff(length[seg]) <= 0).all()
    bal = balanced_xcd_windows(blk_offsets, n).cpu().numpy()
    per = np.add.reduceat(nst, bal[:8].clip(max=w_total - 1)) if w_total else np.zeros(8)
    assert per.max() <= 1.05 * per.mean() + nst.max()


def test_panel_order_over_ranges_of_equal_work(cuda_device):
    from voltrix.hybrid import longest_first_order
    from voltrix.schedule import split_equal_work

    for num_panels in (16, 37, 455, 4100):
        g = torch.Generator().manual_seed(num_panels)
        nks = torch.randint(0, 50, (num_panels,), generator=g) * (torch.arange(num_panels) % 7 == 0).long() * 9 + \
            torch.randint(0, 6, (num_panels,), generator=g)
        panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
        panel_ptr[1:] = nks.cumsum(0)
        xcd_ptr = split_equal_work(nks)
        native = longest_first_order(panel_ptr.cuda(), 1, xcd_ptr.cuda())
        ref = longest_first_order(panel_ptr, 1, xcd_ptr)
        assert native.is_cuda and torch.equal(native.cpu(), ref)
        xp = xcd_ptr.tolist()
        for x in range(8):       # positions [xp[x], xp[x+1]) hold exactly the panels xp[x] .. xp[x+1]-1, longest first
            seg = ref[xp[x]:xp[x + 1]].long()
            assert torch.equal(torch.sort(seg).values, torch.arange(xp[x], xp[x + 1]))
            assert bool((nks[seg][1:] <= nks[seg][:-1]).all())


@pytest.mark.parametrize("group", [1, 4, 7])
@pytest.mark.parametrize("num_panels", [1, 5, 8, 9, 455, 4100])
def test_native_panel_order_equals_the_torch_restatement(cuda_device, num_panels, group):
    from voltrix.hybrid import longest_first_order

    g = torch.Generator().manual_seed(num_panels)
    nks = torch.randint(0, 7 if num_panels > 100 else 1000, (num_panels,), generator=g)   # many ties in the large cases
    panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
    panel_ptr[1:] = nks.cumsum(0)
    native = longest_first_order(panel_ptr.cuda(), group)
    ref = longest_first_order(panel_ptr, group)   # CPU tensor -> torch argsort
    assert native.is_cuda and torch.equal(native.cpu(), ref)
    assert torch.equal(torch.sort(native.cpu().long()).values, torch.arange(num_panels))


@pytest.mark.parametrize("kind", ["f16", "bf16"])
@pytest.mark.parametrize("name,scale,num_feats,max_stages", [("reddit_like", 0.05, 128, None), ("reddit_like", 0.05, 104, 13),
                                                             ("products_like", 0.02, 128, 4), ("powerlaw_4m", 0.004, 72, None),
                                                             ("reddit_like", 0.03, 264, None), ("products_like", 0.01, 512, 5)])
def test_two_units_per_wave_give_the_same_bits(cuda_device, kind, name, scale, num_feats, max_stages):
    """spmm_tc16_pair_kernel (units_per_wave = 2): every wave runs two consecutive units of the table, stages alternating
    through one ring into two accumulator sets.  Same bits as one unit per wave -- store mode and atomic mode onto zeros --
    with odd unit counts per XCD, units of unequal length, empty windows, the N % 16 tail, a partially filled slab, and
    several column slabs (slab-major: a pair never straddles two slabs)."""
    indptr, indices, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n, e = indptr.numel() - 1, indices.numel()
    indptr = indptr.clone()
    # a few empty rows -> empty windows (one all-zero TC block each)
    h = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    tb = unit_table(h[0], n, max_stages)
    dtype = torch.float16 if kind == "f16" else torch.bfloat16
    feat = torch.randn(n, num_feats, device="cuda").to(dtype)
    feat[0] = float("nan")   # row 0 of B is what padded slots and idle stages must never multiply in
    s = torch.cuda.current_stream().cuda_stream
    outs = {}
    for atomic in (False, True):
        for nu in (1, 2):
            out = torch.zeros(n, num_feats, device="cuda") if atomic else torch.full((n, num_feats), 7.0, device="cuda")
            buf = torch.full((max(1, tb.num_slots) * 16 * num_feats,), float("nan"), device="cuda")
            assert capi.launch_spmm_sched(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, e, num_feats, feat.data_ptr(),
                                          out.data_ptr(), (128, 3, 4), s, 0, 0, atomic, kind == "bf16", tb, buf.data_ptr(), 0,
                                          nu) == 0
            assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, num_feats, atomic, s) == 0
            outs[(atomic, nu)] = out
    torch.cuda.synchronize()
    for atomic in (False, True):   # NaN positions and every other bit
        a, b = outs[(atomic, 1)], outs[(atomic, 2)]
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))
    # NaN only where row 0 of B is really referenced
    deg0 = torch.zeros(n, dtype=torch.bool, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (indptr[1:] - indptr[:-1]).long())
    deg0[rows[indices.long() == 0]] = True
    nan_rows = torch.isnan(outs[(False, 2)]).any(dim=1)
    windows_with_nan = torch.zeros((n + 15) // 16, dtype=torch.bool, device="cuda")
    windows_with_nan[(deg0.nonzero().flatten() // 16)] = True
    assert not nan_rows[~windows_with_nan.repeat_interleave(16)[:n]].any()   # NaN stays inside the windows that gather row 0


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 455, 5000, 1 << 20])
def test_native_xcd_ranges_equal_the_torch_restatement(cuda_device, n):
    """voltrix_launch_xcd_ranges_of_work / _of_windows / _of_panels (schedule_tables.hpp) against schedule.split_equal_work and
    hybrid.xcd_ranges_of_panels_torch, element by element: skewed work, zeros, every alignment, the all-zero case (equal
    counts), fewer items than XCDs, a million items (every thread of the one workgroup owns a chunk)."""
    from voltrix import hybrid
    from voltrix.schedule import split_equal_work

    g = torch.Generator().manual_seed(n + 1)
    for kind in ("skewed", "ones", "zeros", "one hot"):
        work = {"skewed": (torch.rand(n, generator=g) ** 6 * 5000).to(torch.int32) * (torch.rand(n, generator=g) < 0.7),
                "ones": torch.ones(n, dtype=torch.int32), "zeros": torch.zeros(n, dtype=torch.int32),
                "one hot": torch.zeros(n, dtype=torch.int32)}[kind].to(torch.int32)
        if kind == "one hot" and n:
            work[n // 3] = 7
        for align in (1, 4, 32):
            native = capi.xcd_ranges_of_work(work.cuda(), align)
            ref = split_equal_work(work.to(torch.int64), align)
            assert torch.equal(native.cpu(), ref), (kind, align, native.tolist(), ref.tolist())
    # windows of a handle: stages = ceil(TC blocks / 4)
    counts = torch.randint(1, 90, (n,), generator=g) * (torch.rand(n, generator=g) < 0.8) + 1
    blk = torch.zeros(n + 1, dtype=torch.int32)
    blk[1:] = counts.cumsum(0)
    num_nodes = max(0, 16 * n - 5)
    for align in (1, 32):
        native = capi.xcd_ranges_of_windows(blk.cuda(), num_nodes, align)
        ref = split_equal_work((counts.to(torch.int64) + 3) // 4, align) if n else torch.zeros(9, dtype=torch.int32)
        assert torch.equal(native.cpu(), ref), (align, native.tolist(), ref.tolist())
    # panels: 32 windows each (the last one partial), k-steps per panel, cost 6.6 and a cost with exact halves (2.5)
    num_panels = (num_nodes + 511) // 512
    if num_panels:
        nks = torch.randint(0, 300, (num_panels,), generator=g) * (torch.rand(num_panels, generator=g) < 0.5)
        panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
        panel_ptr[1:] = nks.cumsum(0)
        for cost in (66, 25, 0):
            native = capi.xcd_ranges_of_panels(panel_ptr.cuda(), blk.cuda(), num_nodes, 512, cost)
            ref = hybrid.xcd_ranges_of_panels_torch(panel_ptr, blk, num_nodes, 512, cost)
            assert torch.equal(native[0].cpu(), ref[0]) and torch.equal(native[1].cpu(), ref[1]), cost


@pytest.mark.parametrize("cap", [1, 7, 64, 10 ** 6])
@pytest.mark.parametrize("num_panels", [1, 9, 200, 4100])
def test_native_piece_table_equals_the_torch_restatement(cuda_device, num_panels, cap):
    """voltrix_launch_panel_parts_count / _fill against hybrid.panel_parts_torch: every array, element by element (the order
    inside an XCD range is longest first, ties by (panel, piece): a function of panel_ptr alone), with and without ranges of
    equal work, panels without k-steps, many equal lengths."""
    from voltrix import hybrid
    from voltrix.schedule import split_equal_work

    if cap == 1 and num_panels > 1000:
        pytest.skip("every k-step its own piece on thousands of panels: a table nobody builds")
    g = torch.Generator().manual_seed(num_panels + cap)
    nks = torch.randint(0, 40, (num_panels,), generator=g) * (torch.arange(num_panels) % 5 == 0).long() * 11 + \
        torch.randint(0, 9, (num_panels,), generator=g)
    panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int32)
    panel_ptr[1:] = nks.cumsum(0)
    for xcd_ptr in (None, split_equal_work(nks), torch.tensor([0, 0, 0, num_panels // 2, num_panels // 2, num_panels, num_panels,
                                                                 num_panels, num_panels], dtype=torch.int32)):
        native = hybrid.panel_parts(panel_ptr.cuda(), cap, xcd_ptr.cuda() if xcd_ptr is not None else None)
        ref = hybrid.panel_parts_torch(panel_ptr, cap, xcd_ptr)
        assert (native.num_parts, native.num_cuts, native.num_slots, native.max_parts_per_xcd, native.cap) == \
            (ref.num_parts, ref.num_cuts, ref.num_slots, ref.max_parts_per_xcd, ref.cap)
        assert torch.equal(native.xcd_ptr.cpu(), ref.xcd_ptr)
        assert torch.equal(native.cuts.cpu(), ref.cuts)
        assert torch.equal(native.parts.cpu(), ref.parts)


# ---- round 5: the stream kernel's tables (voltrix/stream_table.hpp behind voltrix_launch_stream_table_count / _fill) -----------
@pytest.mark.parametrize("name,scale", [("yeast_like", 0.02), ("dd_like", 0.2), ("web_berkstan_like", 0.1), ("reddit_like", 0.02),
                                        ("powerlaw_4m", 0.002)])
@pytest.mark.parametrize("run_cost,cut", [(None, None), (6, 6), (48, 3), (2, 1), (128, 100000)])
def test_stream_tables_of_the_library_match_their_restatement(cuda_device, name, scale, run_cost, cut):
    """Element by element: units, runs, XCD ranges, cuts and the header figures of the device builder against
    schedule.stream_tables_torch (graphs of one-stage windows, of hub windows, of long windows; forced and default bounds)."""
    import synth_graphs
    import voltrix
    from voltrix.schedule import stream_tables, stream_tables_torch

    indptr, indices, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n = indptr.numel() - 1
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    got = stream_tables(*handle, n, run_cost=run_cost, cut_stages=cut)
    want = stream_tables_torch(*handle, n, run_cost=run_cost, cut_stages=cut)
    for field in ("run_cost", "cut_stages", "num_units", "num_runs", "num_cuts", "num_slots", "max_runs_per_xcd"):
        assert getattr(got, field) == getattr(want, field), field
    for field in ("units", "runs", "run_ptr", "cuts"):
        assert torch.equal(getattr(got, field), getattr(want, field)), field
    assert int(got.runs[:, 1].max()) <= 64


def test_stream_kernel_through_the_c_abi(cuda_device):
    """voltrix_launch_stream_table_count / _fill + voltrix_launch_spmm_stream_f16 / _bf16 + voltrix_launch_combine_partials, the
    sequence a C host runs: exact on integer operands against the oracle, every ahead-of-time tile, default tile included."""
    import synth_graphs
    import voltrix
    from oracle import torch_ref
    from voltrix import capi
    from voltrix.schedule import stream_tables

    indptr, indices, _ = synth_graphs.generate("web_berkstan_like", scale=0.05)     # hub windows: cuts and partial tiles
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    table = stream_tables(*handle, n)
    assert table.num_cuts > 0
    for dtype in (torch.float16, torch.bfloat16):
        for f, tiles in ((136, [(0, 0, 0), (128, 2, 1), (128, 3, 2), (128, 4, 1), (64, 2, 2)]), (32, [(0, 0, 0), (32, 4, 1)])):
            feat = torch.randint(-3, 4, (n, f)).to(dtype)
            ref = torch_ref.spmm(indptr.numpy(), indices.numpy(), feat.float(), n)
            dev_feat = feat.cuda()
            for tile in tiles:
                out = torch.full((n, f), float("nan"), device="cuda")
                partials = torch.empty(max(1, table.num_slots) * 16 * f, device="cuda")
                assert capi.launch_spmm_stream(handle[1], handle[2], n, f, dev_feat, out, table, partials, tile=tile) == 0
                rc = capi.launch_combine_partials(table, partials.data_ptr(), out.data_ptr(), n, f, False,
                                                  torch.cuda.current_stream().cuda_stream)
                assert rc == 0
                assert torch.equal(out.cpu(), ref), (dtype, f, tile)
    out = torch.empty((n, 128), device="cuda")
    assert capi.launch_spmm_stream(handle[1], handle[2], n, 128, torch.zeros(n, 128, device="cuda").half(), out, table,
                                   torch.empty(16, device="cuda"), tile=(128, 5, 1)) == 3          # not an ahead-of-time tile
