"""GPU: the SpMM hot path against the oracle (torch.sparse.mm on CPU = the reference's own oracle call).

Tolerances (fp32 output; stated per BASELINE.md section 2 / SURVEY.md section 8c):
  fp16 operand (or fp32 operand rounded to fp16 -- the gfx950 stand-in for the reference's TF32 rounding):
      norm-wise  ||out - ref||_2 / ||ref||_2 <= 1e-3   (BASELINE.json asks <= 1e-2)
      calc_diff(out, ref) <= 1e-5                      (the reference's "difference rate 0.00 %")
      element-wise |out - ref|_ij <= (2^-11 + deg_i 2^-23) (A |B|)_ij + deg_i 2^-25   (last term: fp16 subnormals)
  against the oracle evaluated on the SAME rounded operand, and for the exact-fp32 kernel against the fp32 oracle:
      element-wise |out - ref|_ij <= deg_i 2^-23 (A |B|)_ij  (accumulation order only)
"""
import ctypes

import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import synth_graphs
import voltrix
from conftest import load_csr_fixture
from oracle import oracle_np, torch_ref
from voltrix import capi

pytestmark = pytest.mark.gpu


def _bounds(indptr, indices, feat, n, operand_rounded):
    deg = np.diff(np.asarray(indptr, np.int64))[:n].astype(np.float64)
    aabs = oracle_np.spmm_csr(indptr, indices, np.abs(np.asarray(feat, np.float64)), n)
    u = 0.0 if operand_rounded else 2.0 ** -11
    sub = 0.0 if operand_rounded else deg[:, None] * 2.0 ** -25  # fp16 subnormals round with an absolute 2^-25 error
    return (u + deg[:, None] * 2.0 ** -23) * aabs + sub + 1e-30


def _assert_close(out, indptr, indices, feat32, n, mode):
    """mode: 'fp16' (fp16 operand), 'fp16-scaled' (fp32 operand rounded to fp16 after voltrix.spmm's power-of-two
    rescale), 'exact' (fp32 operand, exact products)."""
    out = out.detach().cpu().numpy().astype(np.float64)
    ref = torch_ref.spmm(indptr, indices, feat32, n).numpy().astype(np.float64)
    assert not np.isnan(out).any()
    if mode in ("fp16", "fp16-scaled"):
        ref_same = torch_ref.spmm(indptr, indices, feat32, n, operand_rounding=mode).numpy().astype(np.float64)
        rounded = feat32.half().float() if mode == "fp16" else torch_ref.round_fp16_scaled(feat32)
        assert (np.abs(out - ref_same) <= _bounds(indptr, indices, rounded, n, True)).all()
        assert (np.abs(out - ref) <= _bounds(indptr, indices, feat32, n, False)).all()
        if np.linalg.norm(ref) > 0:
            assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= 1e-3
    else:
        assert (np.abs(out - ref) <= _bounds(indptr, indices, feat32, n, True)).all()
        if np.linalg.norm(ref) > 0:
            assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= 1e-6
    if np.linalg.norm(ref) > 0:
        assert abs(oracle_np.calc_diff(out, ref)) <= 1e-5


@pytest.mark.parametrize("dtype,mode", [(torch.float16, "fp16"), (torch.bfloat16, "exact"), (torch.float32, "fp16-scaled"),
                                        (torch.float32, "exact")])
def test_operator_api_on_fixtures(cuda_device, csr_fixture, dtype, mode, monkeypatch):
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact" if mode == "exact" else "fp16")
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = csr_fixture
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    handle[1].hash_tag = f"fixture_{n}_{len(g['indices'])}"
    feat32 = torch.from_numpy(g["feat"]).float()
    if dtype != torch.float32:
        feat32 = feat32.to(dtype).float()  # the caller's data IS fp16 / bfloat16
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(g["indices"]), feat=feat32.to(dtype).cuda())
    assert out.dtype == torch.float32 and out.shape == (n, feat32.shape[1]) and out.is_cuda
    _assert_close(out, g["indptr"], g["indices"], feat32, n, mode)


def test_reference_test_inputs_with_autotune(cuda_device, monkeypatch):
    """tests/test_spmm.py / test_spmm_kernel.py defaults: N=8192, density 0.01, F=512, seed 20, fp32 feat, hash_tag
    set by the caller; goes through the autotuner like the reference's first call."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "default")
    np.random.seed(20)
    torch.manual_seed(20)
    n, f = 8192, 512
    a = sp.random(n, n, density=0.01, format="csr")
    indptr = torch.tensor(a.indptr, dtype=torch.int32)
    indices = torch.tensor(a.indices, dtype=torch.int32)
    feat = torch.randn(n, f, dtype=torch.float32)
    blk_offsets, hspa_packed, hind = voltrix.csr_preprocess(indptr, indices, n)
    hspa_packed.hash_tag = "test_20_8192_0.01"
    out = voltrix.spmm(blk_offsets, hspa_packed, hind, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda())
    _assert_close(out, a.indptr, a.indices, feat, n, "fp16-scaled")
    ref = torch_ref.spmm(a.indptr, a.indices, feat, n)
    assert float(voltrix.utils.calc_diff(out.cpu(), ref)) * 100 < 1e-3  # "difference rate: 0.000%"
    # second call reuses the tuned kernel and gives the same bits
    out2 = voltrix.spmm(blk_offsets, hspa_packed, hind, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda())
    assert torch.equal(out, out2)


def test_reference_test_spmm_default_case(cuda_device, monkeypatch):
    """The reference's tests/test_spmm.py with ITS OWN defaults (:101-106): seed 20, N = 8192, density 0.1 (6.7 M edges),
    F = 512, fp32 features, hash_tag set by the caller -- on the committed CSR arrays of that input
    (tests/golden/sprandom_N8192_d0.1_seed20.npz; sp.random streams differ across numpy versions).  The reference prints
    "difference rate: 0.000%" against cuSPARSE; the same criterion (calc_diff x 100 rounds to 0.000) against the oracle call
    torch.sparse.mm, plus the element-wise bound of the scaled-fp16 operand."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "default")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sprandom_N8192_d0.1_seed20.npz"))
    n, f = 8192, 512
    dense = np.unpackbits(z["bits"], axis=1)[:, :n].astype(bool)      # rows stored as bitmaps, as tests/test_oracle_goldens.py reads them
    indptr_np = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    indices_np = np.nonzero(dense)[1].astype(np.int32)
    assert len(indices_np) == 6710886 and len(indptr_np) == n + 1           # the nnz SURVEY.md section 8c records for this input
    torch.manual_seed(20)
    feat = torch.randn(n, f, dtype=torch.float32)
    indptr, indices = torch.from_numpy(indptr_np), torch.from_numpy(indices_np)
    blk_offsets, hspa_packed, hind = voltrix.csr_preprocess(indptr, indices, n)
    assert int(blk_offsets[-1]) == 427364                                     # T recorded from the reference's preprocess
    hspa_packed.hash_tag = "test_20_8192_0.1"
    out = voltrix.spmm(blk_offsets, hspa_packed, hind, num_nodes=n, num_edges=indices.numel(), feat=feat.cuda())
    ref = torch_ref.spmm(indptr_np, indices_np, feat, n)
    assert f"{float(voltrix.utils.calc_diff(out.cpu(), ref)) * 100:.3f}" in ("0.000", "-0.000")
    _assert_close(out, indptr_np, indices_np, feat, n, "fp16-scaled")


def _launch(handle, n, e, feat, is_f16, tile, stream=None, prefill=float("nan")):
    out = torch.full((n, feat.shape[1]), prefill, device="cuda")
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, feat.shape[1],
                          feat.data_ptr(), out.data_ptr(), is_f16, tile, s)
    return rc, out


@pytest.mark.parametrize("kind", ["f16", "bf16", "f32"])
def test_every_ahead_of_time_tile_through_the_c_abi(cuda_device, kind):
    g = load_csr_fixture("skewed_1005")  # N % 16 = 13, empty rows, windows from 1 to >100 TC blocks
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(),
                                                 torch.from_numpy(g["indices"]).cuda(), n)[:3]
    feat32 = torch.randn(n, 136)  # 136 = 128 + 8: slab tail for every FS
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[kind]
    feat32 = feat32.to(dtype).float()       # the caller's data IS of the operand type: no operand rounding left
    dev_feat = feat32.to(dtype).cuda()
    tiles = capi.tiles(kind != "f32")
    assert len(tiles) >= 20
    for tile in tiles:
        rc, out = _launch(handle, n, e, dev_feat, {"f16": True, "bf16": "bf16", "f32": False}[kind], tile)
        torch.cuda.synchronize()
        assert rc == 0, tile
        _assert_close(out, g["indptr"], g["indices"], feat32, n, "exact")  # only fp32 accumulation-order error


@pytest.mark.parametrize("num_feats", [8, 24, 32, 40, 64, 100, 128, 200, 256, 512])
def test_feature_widths_including_padding(cuda_device, num_feats, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("cora_like")
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    handle[1].hash_tag = "cora_like"
    feat32 = torch.randn(n, num_feats).half().float()
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(g["indices"]), feat=feat32.half().cuda())
    assert out.shape == (n, num_feats) and out.is_contiguous()
    _assert_close(out, g["indptr"], g["indices"], feat32, n, "fp16")


@pytest.mark.parametrize("magnitude", [1e-30, 1e-10, 1.0, 3e4, 1e10, 1e30])
def test_fp32_features_keep_fp32_range(cuda_device, magnitude, monkeypatch):
    """The reference multiplies in TF32 (fp32's exponent range).  fp32 features far outside fp16's range must neither
    overflow nor flush on the default (fp16 MFMA) path: voltrix.spmm rescales by a power of two per call."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "fp16")
    g = load_csr_fixture("cora_like")
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    handle[1].hash_tag = "cora_like"
    torch.manual_seed(3)
    feat32 = torch.randn(n, 64) * magnitude
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(g["indices"]), feat=feat32.cuda())
    assert torch.isfinite(out).all()
    _assert_close(out, g["indptr"], g["indices"], feat32, n, "fp16-scaled")


def test_fp32_features_with_inf_and_nan_do_not_disturb_the_rescale(cuda_device, monkeypatch):
    """Inf / NaN in an fp32 operand: the power-of-two rescale is skipped (scale 1) and they stay non-finite where
    torch.sparse.mm has them.  Like the reference's mma (A = 0 times B = Inf is NaN), the other rows of a 16-row window
    that gathers such a row of B may turn NaN too; windows that do not reference it are unaffected."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "fp16")
    g = load_csr_fixture("cora_like")
    n = int(g["num_nodes"])
    indptr, cols = np.asarray(g["indptr"]), np.asarray(g["indices"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    handle[1].hash_tag = "cora_like"
    torch.manual_seed(5)
    feat32 = torch.randn(n, 16)
    bad_inf, bad_nan = int(cols[0]), int(cols[-1])
    feat32[bad_inf, 3] = float("inf")
    feat32[bad_nan, 5] = float("nan")
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(cols), feat=feat32.cuda()).cpu()
    ref = torch_ref.spmm(g["indptr"], g["indices"], feat32, n)
    assert not torch.isfinite(out[~torch.isfinite(ref)]).any()          # nothing non-finite became finite
    rows = np.repeat(np.arange(n), np.diff(indptr))
    touched = np.unique(rows[(cols == bad_inf) | (cols == bad_nan)] // 16)  # windows that gather a bad row
    clean = np.ones(n, dtype=bool)
    for w in touched:
        clean[16 * w:16 * w + 16] = False
    assert clean.sum() > n // 2
    assert torch.isfinite(out[clean]).all()
    assert torch.allclose(out[clean], ref[clean], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_operator_is_graph_capturable(cuda_device, dtype, monkeypatch):
    """Launch-bound use (small graphs, many calls): after one warm call (tuner, module load) voltrix.spmm issues only
    stream-ordered work -- cast (amax memset + 2 kernels), SpMM -- so it can be captured in a HIP graph and replayed
    with new feature values in the same buffer."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("cora_like")
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    handle[1].hash_tag = "cora_like"
    torch.manual_seed(11)
    feat = torch.randn(n, 32, device="cuda").to(dtype)
    voltrix.spmm(*handle, num_nodes=n, num_edges=len(g["indices"]), feat=feat)  # warm: tuner + first launch
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = voltrix.spmm(*handle, num_nodes=n, num_edges=len(g["indices"]), feat=feat)
    torch.cuda.current_stream().wait_stream(side)
    for scale in (1.0, -3.0):
        feat.copy_((torch.randn(n, 32, device="cuda") * scale).to(dtype))
        graph.replay()
        torch.cuda.synchronize()
        mode = "fp16" if dtype == torch.float16 else "fp16-scaled"
        _assert_close(out, g["indptr"], g["indices"], feat.float().cpu(), n, mode)


@pytest.mark.parametrize("case", ["cora_like fp32", "two-level fp16", "two-level fp16 wide"])
def test_graphed_operator_replays_either_format(cuda_device, case, monkeypatch):
    """voltrix.GraphedSpMM: the operator captured once, replayed with new features; bit-equal to the eager call, for the
    window format (fp32 features: cast kernels inside the graph) and for the two-level format (two streams, atomics,
    combine pass inside the graph)."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    if case.startswith("cora"):
        g = load_csr_fixture("cora_like")
        n, indptr, indices = int(g["num_nodes"]), torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
        feat = torch.randn(n, 32, device="cuda")
    else:
        monkeypatch.setenv("VOLTRIX_HYBRID", "1")
        ip, ix, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.03)
        n, indptr, indices = ip.numel() - 1, ip.cpu(), ix.cpu()
        # "wide": 320 columns = 2.5 slabs of 128, one launch per slab of either kernel inside the captured graph
        feat = torch.randn(n, 320 if case.endswith("wide") else 128, device="cuda").half()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = f"graphed/{case}"
    if not case.startswith("cora"):
        assert voltrix.two_level_of(handle[1]) is not None
    op = voltrix.GraphedSpMM(*handle, n, indices.numel(), feat)
    for scale in (1.0, -2.5, 0.0):
        x = (torch.randn_like(feat.float()) * scale).to(feat.dtype)
        got = op(x).clone()
        want = voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=x)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    ref = torch_ref.spmm(indptr.numpy(), indices.numpy(), x.float().cpu(), n)
    assert torch.equal(got.cpu(), ref) if scale == 0.0 else True


def test_every_output_row_is_written_and_empty_windows_are_zero(cuda_device):
    g = load_csr_fixture("toy40")  # rows 16..31 empty (one all-zero TC block), N % 16 = 8
    n, e = 40, len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(),
                                                 torch.from_numpy(g["indices"]).cuda(), n)[:3]
    feat = torch.randn(n, 64).half().cuda()
    rc, out = _launch(handle, n, e, feat, True, (64, 4, 1))
    torch.cuda.synchronize()
    assert rc == 0 and not torch.isnan(out).any()
    assert (out[16:32] == 0).all()
    _assert_close(out, g["indptr"], g["indices"], feat.float().cpu(), n, "fp16")


def test_padding_never_gathers_row_zero(cuda_device):
    """Quirk 4 of the reference (unused hind slots = 0 -> B[0] gathered and multiplied by 0 -> NaN/Inf in B[0]
    poisons every padded window).  Here only windows that really reference row 0 may see it."""
    g = load_csr_fixture("skewed_1005")
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(),
                                                 torch.from_numpy(g["indices"]).cuda(), n)[:3]
    feat = torch.randn(n, 32)
    feat[0] = float("nan")
    rows_with_0 = {r for r in range(n) if 0 in g["indices"][g["indptr"][r]:g["indptr"][r + 1]]}
    windows_with_0 = {r // 16 for r in rows_with_0}
    for is_f16, dev_feat, tile in ((True, feat.half().cuda(), (32, 4, 1)), (False, feat.cuda(), (32, 4, 1))):
        rc, out = _launch(handle, n, e, dev_feat, is_f16, tile)
        torch.cuda.synchronize()
        assert rc == 0
        bad_rows = torch.isnan(out).any(dim=1).cpu().numpy().nonzero()[0]
        assert {int(r) // 16 for r in bad_rows} <= windows_with_0
        assert rows_with_0 <= set(int(r) for r in bad_rows)


def test_runs_on_the_callers_stream(cuda_device):
    g = load_csr_fixture("cora_like")
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(),
                                                 torch.from_numpy(g["indices"]).cuda(), n)[:3]
    feat = torch.randn(n, 64).half().cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        rc, out = _launch(handle, n, e, feat, True, (64, 3, 2), stream=side.cuda_stream)
    side.synchronize()
    assert rc == 0
    _assert_close(out, g["indptr"], g["indices"], feat.float().cpu(), n, "fp16")


def _reference_window_order(p1, num_windows, chunk):
    nblk = np.diff(p1)
    wpx = (num_windows + 7) // 8
    order = np.arange(num_windows)
    for x in range(8):
        for b in range(x * wpx, min((x + 1) * wpx, num_windows), chunk):
            e = min(b + chunk, (x + 1) * wpx, num_windows)
            idx = np.arange(b, e)
            order[b:e] = idx[np.lexsort((idx, -nblk[b:e]))]  # descending block count, ties by index
    return order


@pytest.mark.parametrize("chunk", [128, 512, 2048, 1000, 7])
def test_window_order_schedule_is_a_sorted_permutation_and_changes_no_bit(cuda_device, chunk):
    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.15)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    w = (n + 15) // 16
    order = torch.full((w,), -1, dtype=torch.int32, device="cuda")
    capi.launch_window_order(handle[0], n, order, torch.cuda.current_stream().cuda_stream, chunk)
    torch.cuda.synchronize()
    got = order.cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(w))
    assert np.array_equal(got, _reference_window_order(handle[0].cpu().numpy(), w, chunk))
    feat = torch.randn(n, 128, device="cuda").half()
    out0 = torch.full((n, 128), float("nan"), device="cuda")
    out1 = torch.full((n, 128), float("nan"), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for out, ordptr in ((out0, 0), (out1, order.data_ptr())):
        rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, 128,
                              feat.data_ptr(), out.data_ptr(), True, (64, 4, 4), s, ordptr)
        assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out0, out1) and not torch.isnan(out0).any()


@pytest.mark.parametrize("kind,num_feats,tile", [("f16", 264, (128, 3, 4)), ("f16", 512, (64, 3, 4)), ("bf16", 320, (128, 3, 4)),
                                                 ("f16", 96, (32, 4, 4)), ("f32", 200, (64, 3, 1))])
def test_several_column_slabs_with_and_without_a_unit_table(cuda_device, kind, num_feats, tile, monkeypatch):
    """F > FS: the unit order over (window, slab) the launcher picks (spmm_kernels.hpp::slab_major_order: slab-major from
    128-byte row pieces on, window-major below) gives the same bits with the natural window order and with a unit table --
    every output element has the same addends in the same order; last slab partially filled in every case."""
    from voltrix.schedule import unit_table

    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.05)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    tb = unit_table(handle[0], n)
    assert tb.num_cuts > 0
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[kind]
    feat = torch.randn(n, num_feats, device="cuda").to(dtype)
    s = torch.cuda.current_stream().cuda_stream
    outs = {}
    for order in ("rule",):
        out = torch.full((n, num_feats), float("nan"), device="cuda")
        if kind == "f32":
            rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, num_feats,
                                  feat.data_ptr(), out.data_ptr(), False, tile, s)
            assert rc == 0
        else:
            rc = capi.launch_spmm_sched(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, num_feats,
                                        feat.data_ptr(), out.data_ptr(), tile, s, bf16=kind == "bf16")
            assert rc == 0
            out_t = torch.full((n, num_feats), float("nan"), device="cuda")
            buf = torch.empty(max(1, tb.num_slots) * 16 * num_feats, dtype=torch.float32, device="cuda")
            assert capi.launch_spmm_sched(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, num_feats,
                                          feat.data_ptr(), out_t.data_ptr(), tile, s, 0, 0, False, kind == "bf16", tb,
                                          buf.data_ptr()) == 0
            assert capi.launch_combine_partials(tb, buf.data_ptr(), out_t.data_ptr(), n, num_feats, False, s) == 0
            outs[(order, "table")] = out_t
        outs[(order, "natural")] = out
    torch.cuda.synchronize()
    ref = torch_ref.spmm(indptr.cpu().numpy(), indices.cpu().numpy(), feat.float().cpu(), n)
    for key, out in outs.items():
        assert not torch.isnan(out).any(), key
        got = out.cpu()            # the unit table sums the cut windows' partial tiles in unit order: another fp32 order
        assert torch.linalg.norm(got - ref) / torch.linalg.norm(ref) <= (1e-3 if kind != "f32" else 1e-6), key
    if ("rule", "table") in outs:
        a, b = outs[("rule", "table")], outs[("rule", "natural")]
        assert float((a - b).norm() / b.norm()) < 1e-5


def test_cast_entry_point(cuda_device):
    x = torch.randn(1000, 64, device="cuda") * 300
    y = torch.empty(1000, 64, dtype=torch.float16, device="cuda")
    capi.launch_cast_f32_f16(x, y, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(y, x.half())


@pytest.mark.parametrize("name,scale,num_feats", [("reddit_like", 1.0, 128), ("products_like", 0.25, 512)])
def test_full_size_properties(cuda_device, name, scale, num_feats):
    """BASELINE.json sizes, size-independent properties instead of a CPU oracle:
    (1) A @ ones == distinct degree, exactly (sums of 1.0 up to 21,657 are exact in fp32);
    (2) linearity: A @ (x + y) == A @ x + A @ y up to accumulation-order error;
    (3) checksum of checksums: column sums of A @ x == (A^T 1) . x computed from the degree of each COLUMN."""
    indptr, indices, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    tile = capi.default_tile(num_feats, True)
    ones = torch.ones(n, num_feats, dtype=torch.float16, device="cuda")
    rc, out = _launch(handle, n, e, ones, True, tile)
    assert rc == 0
    deg = (indptr[1:] - indptr[:-1]).float()
    assert torch.equal(out, deg[:, None].expand(n, num_feats))
    del ones, out
    # integer-valued operands keep every partial sum exact -> bit-exact linearity and checksum
    x = torch.randint(-3, 4, (n, num_feats), device="cuda").half()
    y = torch.randint(-3, 4, (n, num_feats), device="cuda").half()
    _, ox = _launch(handle, n, e, x, True, tile)
    _, oy = _launch(handle, n, e, y, True, tile)
    _, oxy = _launch(handle, n, e, x + y, True, tile)
    assert torch.equal(oxy, ox + oy)
    col_deg = torch.bincount(indices.long(), minlength=n).double()
    assert torch.equal(ox.double().sum(dim=0), (col_deg[:, None] * x.double()).sum(dim=0))


def test_route_b_fp32_entry_on_the_fp16_path(cuda_device):
    """voltrix_launch_spmm_f32_as_f16: the reference's fp32 launch arguments + a caller-owned workspace; magnitudes far
    outside fp16's range survive (one power-of-two rescale per call), the mantissa is the reference's TF32 one."""
    g = load_csr_fixture("skewed_1005")
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_fused_preprocess_kernel(torch.from_numpy(g["indptr"]).cuda(), torch.from_numpy(g["indices"]).cuda(), n)[:3]
    for scale in (1.0, 3e7, 2e-9):
        feat32 = torch.randn(n, 64) * scale
        feat = feat32.cuda()
        ws = torch.empty(capi.spmm_f32_workspace_bytes(n, 64), dtype=torch.uint8, device="cuda")
        assert ws.numel() == 16 + n * 64 * 2
        out = torch.full((n, 64), float("nan"), device="cuda")
        rc = capi.launch_spmm_f32_as_f16(handle[0], handle[1], handle[2], n, e, 64, feat, out, ws,
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        _assert_close(out, g["indptr"], g["indices"], feat32, n, "fp16-scaled")
    rc = capi.launch_spmm_f32_as_f16(handle[0], handle[1], handle[2], n, e, 60, feat, out, ws,
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 1   # embedding_dim % 8 != 0


def test_kernel_isolated_timing_hook(cuda_device, monkeypatch):
    """utils.bench_kineto (reference utils.py:232-321): the launches of one multi-kernel operator call timed separately --
    cast, window kernel (JIT runtime), panel kernel on its side stream, zero fill."""
    from voltrix.utils import KernelTimer, bench_kineto

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n = indptr.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr, indices, n, tau=130)
    two.hash_tag = "timing_hook"
    feat = torch.randn(n, 128, device="cuda")          # fp32: cast + scaled fp16 path
    for _ in range(2):      # means of five launches of ~50 us each: one stall of the box (seen once: 89 ms) and the mean is off
        t_spmm, t_panel = bench_kineto(lambda: voltrix.spmm_two_level(two, feat), ("spmm_kernel", "spmm_panel"), num_tests=5)
        if t_spmm < 1e-2 and t_panel < 1e-2:
            break
    assert 0 < t_spmm < 1e-2 and 0 < t_panel < 1e-2    # seconds, like the reference
    with KernelTimer() as timer:
        voltrix.spmm_two_level(two, feat)
    names = set(timer.summary())
    assert {"spmm_kernel", "spmm_panel", "cast_f32_f16_scaled", "zero_fill"} <= names, names
    with pytest.raises(AssertionError):
        bench_kineto(lambda: voltrix.spmm_two_level(two, feat), "no_such_kernel", num_tests=1)
