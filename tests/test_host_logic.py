"""CPU: host-side logic of the product package -- operator-API assertions, tile space, tuner key, synthetic graph
generators, roofline accounting, row-window sharding."""
import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from voltrix import dist as vdist
from voltrix.jit_kernels import spmm as spmm_mod


def test_csr_preprocess_asserts_like_reference():
    indptr = torch.tensor([0, 1, 2], dtype=torch.int32)
    with pytest.raises(AssertionError):  # reference spmm.py:21-22: CPU int32 only
        voltrix.csr_preprocess(indptr.long(), torch.tensor([0, 1], dtype=torch.int32), 2)
    with pytest.raises(AssertionError):
        voltrix.csr_preprocess(indptr, torch.tensor([0, 1], dtype=torch.int64), 2)


def test_spmm_kernel_asserts_like_reference():
    t = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(AssertionError):  # reference jit_kernels/spmm.py:50-54: CUDA tensors required
        voltrix.spmm_kernel(t, t, t, num_nodes=1, num_edges=0, embedding_dim=8, input=torch.zeros(1, 8),
                            output=torch.zeros(1, 8))
    with pytest.raises(AssertionError):
        voltrix.spmm(t, t, t, 1, 0, torch.zeros(1, 8))


def test_tile_space_is_valid_and_bounded(monkeypatch):
    for mode, lo, hi in (("default", 16, 74), ("full", 36, 226), ("none", 1, 1), ("stream", 1, 1)):  # fp32: + 4 stream points
        monkeypatch.setenv("VOLTRIX_TUNE_SPACE", mode)
        for f in (16, 32, 64, 128, 512):
            for eb in (2, 4):
                space = spmm_mod.tile_space(f, eb)
                assert lo <= len(space) <= hi, (mode, f, eb, len(space))
                for p in space:
                    assert spmm_mod._lds_bytes(p["FS"], p["DEPTH"], p["WAVES"], p["EB"]) <= 160 * 1024
                    assert p["EB"] == eb and p["FS"] in (32, 64, 128, 256) and p["SCHED"] in ((0, 1, 2, 3, 4, 5, 6) if eb == 2 else (0, 1, 2, 3, 6))
                    assert p["SCHED"] != 5 or (p["WAVES"] == 4 and (p["FS"] >= 64 or f <= p["FS"]))   # paired units
                    if p["SCHED"] == 6:   # stream points: loads + stores behind a counted wait fit the 6-bit vmcnt
                        ndma, slots = 32 * p["FS"] * eb // 1024, p["FS"] // 16
                        assert eb == 2 or p["FS"] <= 64
                        assert p["WAVES"] in (1, 2) and (1 + ndma) * (p["DEPTH"] - 1) + p["DEPTH"] * slots <= 63
                # the stream kernel serves plain stores of a binary 16-bit operand only
                for kw in (dict(weighted=True), dict(max_lds=spmm_mod.TWO_LEVEL_LDS_BUDGET), dict(stream_ok=False)):
                    if eb == 2:
                        assert all(p["SCHED"] != 6 for p in spmm_mod.tile_space(f, eb, **kw)), (mode, f, kw)
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    # the single default point = the ahead-of-time library's default tile + the unit-table schedule
    assert spmm_mod.tile_space(128, 2) == ({"FS": 128, "DEPTH": 3, "WAVES": 4, "EB": 2, "SCHED": 4, "BF16": 0, "WEIGHTED": 0},)
    assert spmm_mod.tile_space(128, 2, bf16=True) == ({"FS": 128, "DEPTH": 3, "WAVES": 4, "EB": 2, "SCHED": 4, "BF16": 1, "WEIGHTED": 0},)
    assert spmm_mod.tile_space(64, 2) == ({"FS": 64, "DEPTH": 3, "WAVES": 4, "EB": 2, "SCHED": 4, "BF16": 0, "WEIGHTED": 0},)
    # ... beside a panel workgroup (two-level format): two units per wave
    assert spmm_mod.tile_space(128, 2, max_lds=spmm_mod.TWO_LEVEL_LDS_BUDGET) == (
        {"FS": 128, "DEPTH": 3, "WAVES": 4, "EB": 2, "SCHED": 5, "BF16": 0, "WEIGHTED": 0},)
    assert spmm_mod.tile_space(128, 2, weighted=True) == ({"FS": 128, "DEPTH": 3, "WAVES": 4, "EB": 2, "SCHED": 4, "BF16": 0, "WEIGHTED": 1},)
    assert spmm_mod.tile_space(16, 4) == ({"FS": 32, "DEPTH": 3, "WAVES": 1, "EB": 4, "SCHED": 2, "BF16": 0, "WEIGHTED": 0},)


def test_feature_hash_uses_tag_then_address():
    t = torch.zeros(4, dtype=torch.int32)
    with pytest.warns(UserWarning, match="hash_tag"):
        h_addr = spmm_mod.feature_hash(t)
    t.hash_tag = "test_20_8192_0.1"
    assert spmm_mod.feature_hash(t) == voltrix.jit.hash_to_hex("test_20_8192_0.1") != h_addr


def test_synthetic_graphs_are_canonical_and_seeded():
    a = synth_graphs.generate("reddit_like", scale=0.004)
    b = synth_graphs.generate("reddit_like", scale=0.004)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    indptr, indices, cfg = a
    n = indptr.numel() - 1
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and int(indptr[-1]) == indices.numel()
    ip, ix = indptr.numpy(), indices.numpy()
    for r in np.random.default_rng(0).choice(n, 50, replace=False):
        row = ix[ip[r]:ip[r + 1]]
        assert (np.diff(row) > 0).all() and row.min(initial=0) >= 0 and row.max(initial=0) < n
    assert (np.diff(ip) >= 1).all()
    # the published shapes of the BASELINE configs
    assert synth_graphs.CONFIGS["reddit_like"]["num_nodes"] == 232965
    assert synth_graphs.algorithmic_bytes(232965, 114615892, 128, 2) == 4 * (114615892 + 232966) + 232965 * 128 * 6
    assert synth_graphs.flops(114615892, 128) == 2 * 114615892 * 128


def test_partition_rows_balances_edges_on_window_boundaries():
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.01)
    n = indptr.numel() - 1
    for world in (1, 2, 3, 8):
        parts = vdist.partition_rows(indptr, n, world)
        assert parts[0][0] == 0 and parts[-1][1] == n and len(parts) == world
        edges = []
        for (r0, r1), nxt in zip(parts, parts[1:] + [(n, n)]):
            assert r1 == nxt[0] and r0 % 16 == 0 and r0 <= r1
            edges.append(int(indptr[r1]) - int(indptr[r0]))
        assert sum(edges) == indices.numel()
        if world > 1:
            assert max(edges) <= 1.25 * (indices.numel() / world) + 16 * 3000


def test_remap_columns_addresses_the_padded_gather_buffer():
    parts = [(0, 32), (32, 48), (48, 100)]
    rows_padded = 52
    cols = torch.tensor([0, 31, 32, 47, 48, 99], dtype=torch.int32)
    got = vdist.remap_columns(cols, parts, rows_padded).tolist()
    assert got == [0, 31, 52, 52 + 15, 104, 104 + 51]


def test_tile_space_for_two_level_handles_leaves_room_for_the_panel_kernel():
    """Handles of the two-level format run beside the panel kernel: the tuner only sees window tiles whose workgroup fits
    the LDS the panel workgroup (44 KB) leaves on a CU."""
    from voltrix.jit_kernels import spmm as spmm_mod

    full = spmm_mod.tile_space(128, 2)
    fit = spmm_mod.tile_space(128, 2, max_lds=spmm_mod.TWO_LEVEL_LDS_BUDGET)
    assert fit and set(map(lambda p: tuple(sorted(p.items())), fit)) <= set(map(lambda p: tuple(sorted(p.items())), full))
    for p in fit:
        assert spmm_mod._lds_bytes(p["FS"], p["DEPTH"], p["WAVES"], p["EB"]) <= spmm_mod.TWO_LEVEL_LDS_BUDGET
        assert p["WAVES"] >= 4
    assert any(p["FS"] == 128 and p["DEPTH"] == 3 and p["WAVES"] == 4 for p in fit)


def test_every_default_panel_tile_is_instantiated():
    """voltrix.hybrid.default_panel_tile must only name tiles the ahead-of-time library instantiates
    (csrc/capi_spmm_panel.hip), for every plan geometry and feature width."""
    import os
    import re

    from conftest import PKG_ROOT
    from voltrix import hybrid

    src = open(os.path.join(PKG_ROOT, "csrc", "capi_spmm_panel.hip")).read()
    inst = {tuple(map(int, m)) for m in re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+)\)", src)}
    assert len(inst) > 10
    # tiles of the software-pipelined loop: X(FS, DEPTH, WAVES, RB) under ksteps = KSTEPS_PIPELINED
    pipe_space = re.search(r"#define VOLTRIX_PANEL_PIPE_SPACE\(X\)(.*)", src).group(1)
    piped = {tuple(map(int, m)) + (hybrid.KSTEPS_PIPELINED,) for m in re.findall(r"X\((\d+), (\d+), (\d+), (\d+)\)", pipe_space)}
    assert len(piped) >= 4 and f"kPipelined = {hybrid.KSTEPS_PIPELINED};" in src
    inst |= piped
    for feat in (8, 32, 48, 64, 96, 128, 200, 512):
        for waves in (4, 8):
            for rb in (2, 4):
                fs, depth, ksteps = hybrid.default_panel_tile(feat, waves, rb)
                assert (fs, depth, waves, rb, ksteps) in inst, (feat, waves, rb)
