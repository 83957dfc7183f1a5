"""CPU: the stage-list schedule builder (voltrix/schedule.py) -- every TC block is scheduled exactly once, in stages of
<= 4 consecutive blocks of one window; accumulator sets are flushed exactly once per window; lists carry their padding;
a numpy emulation of the executor on the lists reproduces the oracle's SpMM.  GPU: the executor itself."""
import numpy as np
import pytest
import torch

import synth_graphs
from conftest import load_csr_fixture
from oracle import oracle_c, oracle_np
from voltrix.schedule import build_stage_list


def _handle(indptr, indices, n):
    p1, packed, hind = oracle_c.csr_preprocess(np.asarray(indptr, np.int32), np.asarray(indices, np.int32), n)
    return p1, packed, hind


def _emulate(sl, p1, packed, hind, n, feat):
    tiles = oracle_np.unpack_swizzled(packed).astype(np.float64)
    hind2 = np.asarray(hind, np.int64).reshape(-1, 8)
    out = np.full((len(p1) - 1) * 16, np.nan)[:, None] * np.ones((1, feat.shape[1]))
    ent = sl.entries.numpy()
    wp = sl.wave_ptr.numpy()
    seen_blocks = np.zeros(int(p1[-1]), dtype=np.int64)
    flushed = np.zeros(len(p1) - 1, dtype=np.int64)
    pad = 2 * sl.depth + 1
    for v in range(sl.num_waves):
        lst = ent[wp[v]:wp[v + 1]]
        assert len(lst) >= pad and ((lst[-pad:, 1] & 0xFF) == 0).all()  # padding: count 0 (tail bit set)
        acc = np.zeros((sl.groups, 16, feat.shape[1]))
        owner = [-1] * sl.groups
        for b0, info, w, _ in lst[:-pad] if pad else lst:
            cnt, g, fl, tail = info & 0xFF, (info >> 8) & 0xFF, (info >> 16) & 1, (info >> 17) & 1
            assert 0 <= cnt <= 4 and g < sl.groups
            # stages that can hold padded hind slots (window's last block / partial / empty) must take the tail path
            assert tail or (cnt == 4 and b0 + 4 < p1[w + 1])
            assert owner[g] in (-1, w), "accumulator set shared by two live windows"
            owner[g] = w
            if cnt:
                assert p1[w] <= b0 and b0 + cnt <= p1[w + 1]
            for b in range(b0, b0 + cnt):
                seen_blocks[b] += 1
                acc[g] += tiles[b] @ feat[hind2[b]]
            if fl:
                out[w * 16:(w + 1) * 16] = acc[g]
                acc[g] = 0
                owner[g] = -1
                flushed[w] += 1
        assert all(o == -1 for o in owner)
    return out[:n], seen_blocks, flushed


@pytest.mark.parametrize("mode,groups,num_waves", [("plain", 1, 8), ("plain", 4, 16), ("sweep", 4, 8), ("sweep", 8, 24),
                                                  ("sweep", 2, 64)])
@pytest.mark.parametrize("fixture", ["toy40", "skewed_1005", "cora_like"])
def test_stage_lists_cover_everything_once_and_reproduce_the_oracle(fixture, mode, groups, num_waves):
    g = load_csr_fixture(fixture)
    n = int(g["num_nodes"])
    p1, packed, hind = g["pointer1"], g["hspa_packed"], g["hind"]
    sl = build_stage_list(torch.from_numpy(p1), torch.from_numpy(packed.view(np.int32)).view(torch.uint32),
                          torch.from_numpy(hind), n, num_waves=num_waves, groups=groups, depth=3, mode=mode,
                          panel_rows=300, near_rows=100)
    feat = g["feat"].astype(np.float64)
    out, seen, flushed = _emulate(sl, p1, packed, hind, n, feat)
    empty_blocks = np.zeros(int(p1[-1]), dtype=bool)
    for w in range(len(p1) - 1):
        if p1[w + 1] - p1[w] == 1 and not packed[4 * p1[w]:4 * p1[w] + 4].any():
            empty_blocks[p1[w]] = True
    assert (seen[~empty_blocks] == 1).all() and (seen[empty_blocks] == 0).all()
    assert (flushed == 1).all()
    ref = oracle_np.spmm_csr(g["indptr"], g["indices"], feat, n)
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12)
    assert sl.wave_ptr.dtype == torch.int32 and sl.entries.dtype == torch.int32 and sl.entries.shape[1] == 4


def test_sweep_order_is_near_then_panel_major():
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n = indptr.numel() - 1
    p1, packed, hind = _handle(indptr.numpy(), indices.numpy(), n)
    sl = build_stage_list(torch.from_numpy(p1), torch.from_numpy(packed.view(np.int32)).view(torch.uint32),
                          torch.from_numpy(hind), n, num_waves=8, groups=4, depth=4, mode="sweep", panel_rows=512,
                          near_rows=256)
    ent, wp = sl.entries.numpy(), sl.wave_ptr.numpy()
    first_col = hind.reshape(-1, 8)[:, 0]
    lst = ent[wp[0]:wp[1] - 9]
    w0 = lst[:, 2]
    near = np.abs(first_col[lst[:, 0]] - (w0 * 16 + 8)) <= 256
    # inside the first round of wave 0: all near stages come first, then panels ascend
    first_round = np.isin(w0, np.unique(w0)[:4])
    fr_near = near[first_round]
    k = int(fr_near.sum())
    assert fr_near[:k].all() and not fr_near[k:].any()
    panels = first_col[lst[first_round][k:, 0]] // 512
    assert (np.diff(panels) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("groups,depth,fs", [(1, 3, 128), (4, 4, 128), (8, 3, 64), (2, 3, 64)])
def test_executor_plain_schedule_is_bit_identical_to_the_window_kernel(cuda_device, groups, depth, fs):
    import voltrix
    from voltrix import capi

    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.05)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    feat = torch.randn(n, 128, device="cuda").half()
    s = torch.cuda.current_stream().cuda_stream
    ref = torch.full((n, 128), float("nan"), device="cuda")
    assert capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, 128,
                            feat.data_ptr(), ref.data_ptr(), True, (fs, depth, 1), s) == 0
    sl = build_stage_list(handle[0], handle[1], handle[2], n, num_waves=256, groups=groups, depth=depth, mode="plain")
    out = torch.full((n, 128), float("nan"), device="cuda")
    assert capi.launch_spmm_list(handle[1].data_ptr(), handle[2].data_ptr(), n, 128, feat.data_ptr(), out.data_ptr(),
                                 sl.entries, sl.wave_ptr, sl.num_waves, (fs, depth, groups), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.gpu
def test_executor_sweep_schedule_matches_oracle(cuda_device):
    import voltrix
    from oracle import torch_ref
    from voltrix import capi

    g = load_csr_fixture("skewed_1005")
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(),
                                                 torch.from_numpy(g["indices"]).cuda(), n)[:3]
    feat = torch.randint(-3, 4, (n, 128)).half()
    feat[0] = float("nan")  # padding / empty windows must not pull row 0 in
    ref = torch_ref.spmm(g["indptr"], g["indices"], feat.float(), n)
    s = torch.cuda.current_stream().cuda_stream
    dev_feat = feat.cuda()
    windows_with_0 = {r // 16 for r in range(n) if 0 in g["indices"][g["indptr"][r]:g["indptr"][r + 1]]}
    clean = torch.tensor([(r // 16) not in windows_with_0 for r in range(n)])
    for groups, num_waves in ((4, 8), (2, 64), (8, 16)):
        sl = build_stage_list(handle[0], handle[1], handle[2], n, num_waves=num_waves, groups=groups, depth=3,
                              mode="sweep", panel_rows=128, near_rows=64)
        out = torch.full((n, 128), float("nan"), device="cuda")
        assert capi.launch_spmm_list(handle[1].data_ptr(), handle[2].data_ptr(), n, 128, dev_feat.data_ptr(),
                                     out.data_ptr(), sl.entries, sl.wave_ptr, sl.num_waves, (128, 3, groups), s) == 0
        torch.cuda.synchronize()
        # integer-valued operand: exact in any order.  A NaN row of B reaches all 16 rows of a window that references
        # it (0 * NaN in the matrix core, as in the reference's formulation); every other window must be exact.
        assert torch.equal(out.cpu()[clean], ref[clean])
        bad_windows = {int(r) // 16 for r in torch.isnan(out).any(dim=1).cpu().numpy().nonzero()[0]}
        assert bad_windows <= windows_with_0
