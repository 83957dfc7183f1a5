"""GPU: two-level format (panel kernel + window kernel) against the oracle.

The numerical oracle is the reference's own: ``csr(ones) @ feat`` on the CPU (torch_ref.spmm) for the FULL matrix; the
panel kernel alone is checked against the same oracle on the edges the plan's consumer-side interpreter
(oracle_np.panel_to_edges) reads out of the plan.  Tolerances as in test_gpu_spmm.py (fp16 operand: element-wise
(2^-11 + deg 2^-23)(A|B|) + deg 2^-25, norm-wise 1e-3; same-rounded operand: deg 2^-23 (A|B|))."""
import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from oracle import oracle_np, torch_ref
from test_gpu_spmm import _assert_close
from test_hybrid_plan import _random_csr
from voltrix import hybrid

pytestmark = pytest.mark.gpu

# (waves, row_blocks, (fs, depth, ksteps), F)
PANEL_CASES = [
    (8, 4, (128, 6, 1), 128), (8, 4, (128, 4, 1), 256), (8, 4, (128, 8, 1), 128), (4, 4, (128, 6, 1), 128),
    (8, 2, (128, 6, 1), 128), (4, 2, (128, 6, 1), 384), (4, 4, (64, 6, 1), 64), (8, 4, (64, 6, 2), 64),
    (4, 4, (64, 6, 2), 128), (4, 4, (32, 6, 2), 32), (8, 4, (32, 6, 2), 32), (8, 4, (128, 6, 1), 96),
    # the software-pipelined loop (ksteps = hybrid.KSTEPS_PIPELINED; round 5)
    (8, 4, (128, 3, 17), 128), (8, 4, (128, 4, 17), 200), (8, 2, (128, 4, 17), 128), (8, 2, (128, 6, 17), 384),
    (8, 4, (64, 4, 17), 64), (8, 4, (64, 6, 17), 72),
]


def _plan_arrays(plan):
    return (plan.panel_ptr.cpu().numpy(), plan.panel_cols.cpu().numpy(),
            plan.panel_bits.view(torch.int32).cpu().numpy().view(np.uint32))


def _check_hip_builder(indptr, indices, n, waves, rb, tau, ncols=None):
    """HIP builder (panel_plan.hpp) == plain-loop oracle, bit for bit."""
    ri, rx, plan = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), n, ncols,
                                           waves, rb, tau)
    o_ri, o_rx, o_ptr, o_cols, o_bits = oracle_np.panel_plan(indptr, indices, n, waves, rb, tau)
    p_ptr, p_cols, p_bits = _plan_arrays(plan)
    assert np.array_equal(ri.cpu().numpy(), o_ri) and np.array_equal(rx.cpu().numpy(), o_rx)
    assert np.array_equal(p_ptr, o_ptr) and np.array_equal(p_cols, o_cols) and np.array_equal(p_bits, o_bits)
    assert plan.num_ksteps == int(o_ptr[-1]) and plan.num_resid_edges == len(o_rx)
    assert plan.num_shared_edges + plan.num_resid_edges == len(np.unique(
        np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr)) * (ncols or n) + indices))
    return plan


@pytest.mark.parametrize("waves,rb", [(4, 2), (8, 4), (4, 4), (8, 2)])
@pytest.mark.parametrize("tau", [1, 2, 4, 40])
def test_hip_plan_builder_matches_oracle(cuda_device, waves, rb, tau):
    indptr, indices = _random_csr(700, 60, seed=waves * 10 + rb + tau)
    _check_hip_builder(indptr, indices, 700, waves, rb, tau)


def test_hip_plan_builder_edge_cases(cuda_device, csr_fixture):
    g = csr_fixture
    _check_hip_builder(g["indptr"], g["indices"], int(g["num_nodes"]), 4, 2, 2)
    indptr, indices = _random_csr(300, 30, seed=5)
    plan = _check_hip_builder(indptr, indices, 300, 4, 2, 60000)        # a threshold nothing reaches
    assert plan.num_ksteps == 0
    _check_hip_builder(np.zeros(41, np.int32), np.zeros(0, np.int32), 40, 4, 2, 2)   # no edges at all
    indptr3, indices3 = _random_csr(200, 40, seed=9, ncols=200000)      # non-square, four column ranges
    _check_hip_builder(indptr3, indices3, 200, 4, 2, 1, ncols=200000)
    wide = np.arange(0, 200000, 7, dtype=np.int32)                      # two rows sharing 28 k columns across ranges
    _check_hip_builder(np.array([0, len(wide), 2 * len(wide)] + [2 * len(wide)] * 30, np.int32), np.r_[wide, wide], 32,
                       4, 2, 2, ncols=200000)
    # unsorted rows with duplicates: detected on the device, canonicalised once, same plan as the clean input
    rng = np.random.default_rng(3)
    rows = [rng.permutation(np.r_[r, r[: len(r) // 3]]) for r in
            (indices[indptr[i]:indptr[i + 1]] for i in range(300))]
    d_indptr = np.zeros(301, np.int32)
    d_indptr[1:] = np.cumsum([len(r) for r in rows])
    ri, rx, plan = hybrid.build_panel_plan(torch.from_numpy(d_indptr).cuda(),
                                           torch.from_numpy(np.concatenate(rows).astype(np.int32)).cuda(), 300, None, 4, 2, 2)
    o = oracle_np.panel_plan(indptr, indices, 300, 4, 2, 2)
    assert np.array_equal(ri.cpu().numpy(), o[0]) and np.array_equal(rx.cpu().numpy(), o[1])
    assert all(np.array_equal(a, b) for a, b in zip(_plan_arrays(plan), o[2:]))
    _, rx_out, plan_out = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), 300,
                                                  100, 4, 2, 2)     # ids beyond the declared universe: no plan
    assert plan_out.num_ksteps == 0 and torch.equal(rx_out.cpu(), torch.from_numpy(indices))


def test_hip_plan_builder_equals_torch_form_at_size(cuda_device):
    """A quarter-size reddit-like graph (27 M edges): HIP builder == torch-op form, every array."""
    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.25)
    n = indptr.numel() - 1
    for waves, rb, tau in ((8, 4, 3), (4, 4, 2)):
        a = hybrid.build_panel_plan(indptr, indices, n, None, waves, rb, tau)
        b = hybrid.build_panel_plan_torch(indptr, indices, n, None, waves, rb, tau)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert torch.equal(a[2].panel_ptr, b[2].panel_ptr) and torch.equal(a[2].panel_cols, b[2].panel_cols)
        assert torch.equal(a[2].panel_bits.view(torch.int32), b[2].panel_bits.view(torch.int32))


def _edges_to_csr(edges, n):
    indptr = np.zeros(n + 1, np.int32)
    for r, _ in edges:
        indptr[r + 1] += 1
    indptr = np.cumsum(indptr).astype(np.int32)
    return indptr, np.asarray([c for _, c in edges], np.int32)


@pytest.mark.parametrize("waves,rb,tile,feat_dim", PANEL_CASES)
def test_panel_kernel_alone(cuda_device, waves, rb, tile, feat_dim):
    n = 1100  # several panels + a partial one; per-panel k-step counts from 0 to > depth
    indptr, indices = _random_csr(n, 90, seed=waves + rb + feat_dim)
    indptr[300:513] = indptr[300]  # a stretch of empty rows is dropped below by rebuilding the CSR
    _, _, plan = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), n, None, waves,
                                         rb, 2)
    shared = oracle_np.panel_to_edges(plan.panel_ptr.cpu().numpy(), plan.panel_cols.cpu().numpy(),
                                      plan.panel_bits.view(torch.int32).cpu().numpy().view(np.uint32), n, waves, rb)
    s_indptr, s_indices = _edges_to_csr(shared, n)
    torch.manual_seed(feat_dim)
    feat = torch.randn(n, feat_dim).half()
    out = torch.full((n, feat_dim), 7.0, dtype=torch.float32, device=cuda_device)
    hybrid.launch_panel(plan, feat.cuda(), out, accumulate=False, tile=tile)     # overwrite: every row written
    _assert_close(out, s_indptr, s_indices, feat.float(), n, "fp16")
    prior = torch.randn(n, feat_dim, device=cuda_device)
    out2 = prior.clone()
    hybrid.launch_panel(plan, feat.cuda(), out2, accumulate=True, tile=tile)     # accumulate: prior + the same product
    out3 = prior.clone()
    hybrid.launch_panel(plan, feat.cuda(), out3, accumulate=2, tile=tile)        # the same by float atomics
    assert torch.equal(out3, out2)
    assert torch.equal(out2, prior + out)


@pytest.mark.parametrize("dtype,mode", [(torch.float16, "fp16"), (torch.bfloat16, "exact"), (torch.float32, "fp16-scaled")])
@pytest.mark.parametrize("feat_dim", [32, 128, 200])
@pytest.mark.parametrize("join", ["atomic", "one-stream"])
def test_hybrid_operator(cuda_device, dtype, mode, feat_dim, join, monkeypatch):
    """Explicit two-level handle through ``spmm_two_level``, the two ways the two halves can meet in C: float atomics
    onto a zero-filled C from two streams (default), or one stream with a read-add-store panel epilogue."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    concurrent = join != "one-stream"
    indptr_t, indices_t, _ = synth_graphs.generate("reddit_like", scale=0.006)
    n = indptr_t.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr_t, indices_t, n, tau=130)  # small graph, dense panels: a high bar
    assert isinstance(two, voltrix.TwoLevelHandle)                        # leaves edges on both sides
    assert two.plan.num_shared_edges > 0 and two.plan.num_resid_edges > 0
    assert two.plan.num_shared_edges + two.plan.num_resid_edges == two.num_edges == indices_t.numel()
    two.hash_tag = f"hybrid_{n}"
    torch.manual_seed(1)
    feat32 = torch.randn(n, feat_dim)
    if dtype != torch.float32:
        feat32 = feat32.to(dtype).float()
    out = voltrix.spmm_two_level(two, feat32.to(dtype).cuda(), concurrent=concurrent)
    assert out.shape == (n, feat_dim) and out.dtype == torch.float32
    _assert_close(out, indptr_t.numpy(), indices_t.numpy(), feat32, n, mode)
    again = voltrix.spmm_two_level(two, feat32.to(dtype).cuda(), concurrent=concurrent)
    assert torch.equal(out, again)   # two addends per element / fixed combine order: run-to-run identical


def test_csr_preprocess_keeps_the_reference_handle_and_attaches_a_sidecar(cuda_device, csr_fixture, monkeypatch):
    """VOLTRIX_HYBRID=1: ``csr_preprocess`` still returns the reference's handle of the WHOLE matrix (bit-exact against the
    oracle); the two-level form rides along as a hint that ``voltrix.spmm`` uses.  Every other consumer of the three
    tensors -- ``spmm_kernel``, a clone of the tensors, the C-ABI -- computes the same product without it."""
    from oracle import oracle_c
    from voltrix import capi

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    g = csr_fixture
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    op1, opacked, ohind = oracle_c.csr_preprocess(g["indptr"], g["indices"], n)
    assert np.array_equal(handle[0].cpu().numpy(), op1) and np.array_equal(handle[2].cpu().numpy(), ohind)
    assert np.array_equal(handle[1].view(torch.int32).cpu().numpy().view(np.uint32), opacked)
    handle[1].hash_tag = f"sidecar_fixture_{n}"
    feat32 = torch.from_numpy(g["feat"]).float().half().float()
    feat = feat32.half().cuda()
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    _assert_close(out, g["indptr"], g["indices"], feat32, n, "fp16")
    if e > 0 and n > 16:
        hint = voltrix.two_level_of(handle[1])
        assert hint is not None and hint.plan.num_shared_edges + hint.plan.num_resid_edges == e
        # the record follows the MEMORY, not the Python object: a view / a re-packed tuple of the handle keeps it
        assert voltrix.two_level_of(handle[1].view(-1)) is hint and voltrix.two_level_of(handle[1].detach()) is hint
    # consumers that never see the side-car: a clone of the tensors, the L3 wrapper, the raw C-ABI launch
    clone = tuple(t.clone() for t in handle)
    clone[1].hash_tag = f"sidecar_clone_{n}"
    assert voltrix.two_level_of(clone[1]) is None and not voltrix.sidecar.lookup(clone[1])[0]   # a COPY of the bytes has no record
    _assert_close(voltrix.spmm(*clone, num_nodes=n, num_edges=e, feat=feat), g["indptr"], g["indices"], feat32, n, "fp16")
    f = feat.shape[1]
    if f % 8 == 0:
        raw = torch.full((n, f), float("nan"), device="cuda")
        voltrix.spmm_kernel(*handle, num_nodes=n, num_edges=e, embedding_dim=f, input=feat, output=raw)
        _assert_close(raw, g["indptr"], g["indices"], feat32, n, "fp16")
        raw2 = torch.full((n, f), float("nan"), device="cuda")
        rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, f, feat.data_ptr(),
                              raw2.data_ptr(), True, capi.default_tile(f, True), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        _assert_close(raw2, g["indptr"], g["indices"], feat32, n, "fp16")
    # VOLTRIX_HYBRID=0: no side-car, same handle bytes
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    plain = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    assert voltrix.sidecar.lookup(plain[1]) == (True, None)      # decided: window format
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(plain, handle))


def test_hybrid_quarter_size_properties(cuda_device, monkeypatch):
    """reddit-like at a quarter of the full size (the full size: tests/test_gpu_full_size.py): A.1 = degree exactly,
    exact on small integers (every partial sum is an integer below 2^24), and the two-level result equals the
    window-format result to accumulation order; also through the side-car of ``csr_preprocess`` (auto mode)."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.25)
    n = indptr.numel() - 1
    indptr_c, indices_c = indptr.cpu(), indices.cpu()
    two = voltrix.csr_preprocess_hybrid(indptr_c, indices_c, n)
    two.hash_tag = "hybrid_quarter"
    assert two.plan.num_shared_edges + two.plan.num_resid_edges == indices.numel()
    ones = torch.ones(n, 128, dtype=torch.float16, device=cuda_device)
    deg = (indptr[1:] - indptr[:-1]).float()
    out = voltrix.spmm_two_level(two, ones)
    assert torch.equal(out, deg[:, None].expand(n, 128))
    ints = torch.randint(-3, 4, (n, 128), device=cuda_device).half()
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    ref_handle = voltrix.csr_preprocess(indptr_c, indices_c, n)
    ref_handle[1].hash_tag = "window_quarter"
    a = voltrix.spmm_two_level(two, ints)
    b = voltrix.spmm(*ref_handle, num_nodes=n, num_edges=indices.numel(), feat=ints)
    assert torch.equal(a, b)
    monkeypatch.setenv("VOLTRIX_HYBRID", "auto")     # 58 k rows = 114 panels: too few to fill 256 CUs, no side-car in auto mode
    auto = voltrix.csr_preprocess(indptr_c, indices_c, n)
    assert voltrix.sidecar.lookup(auto[1]) == (True, None)
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")        # forced: built whenever enough edges sit in shared columns
    forced = voltrix.csr_preprocess(indptr_c, indices_c, n)
    forced[1].hash_tag = "forced_quarter"
    assert voltrix.two_level_of(forced[1]) is not None
    assert all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(forced, ref_handle))
    assert torch.equal(voltrix.spmm(*forced, num_nodes=n, num_edges=indices.numel(), feat=ints), b)


def test_hybrid_operator_is_graph_capturable(cuda_device, monkeypatch):
    """The two-stream form forks and joins through events only: it replays from a HIP graph with the same result."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr_t, indices_t, _ = synth_graphs.generate("reddit_like", scale=0.006)
    n = indptr_t.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr_t, indices_t, n, tau=130)
    two.hash_tag = f"hybrid_graph_{n}"
    feat = torch.randn(n, 128, device=cuda_device).half()
    eager = voltrix.spmm_two_level(two, feat)   # warm-up: tuner, streams, unit table
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = voltrix.spmm_two_level(two, feat)
    feat.copy_(torch.randn(n, 128, device=cuda_device).half())
    graph.replay()
    torch.cuda.synchronize()
    again = voltrix.spmm_two_level(two, feat)
    assert torch.equal(captured, again) and not torch.equal(captured, eager)


def test_hybrid_degenerate_plans(cuda_device, monkeypatch):
    """No shared column at all (threshold out of reach), a column universe above the builder's limit or ids outside the
    declared universe (empty plan by design), a row count that leaves a partial last panel and a feature width that needs
    padding: ``spmm_two_level`` still equals the oracle."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices = _random_csr(777, 50, seed=11)
    feat32 = torch.randn(777, 44).half().float()
    for kwargs in (dict(tau=60000), dict(tau=2, waves=4, row_blocks=2), dict(tau=1)):
        two = voltrix.csr_preprocess_hybrid(torch.from_numpy(indptr), torch.from_numpy(indices), 777, **kwargs)
        two.hash_tag = f"degenerate_{sorted(kwargs.items())}"
        out = voltrix.spmm_two_level(two, feat32.half().cuda())
        _assert_close(out, indptr, indices, feat32, 777, "fp16")
    empty = voltrix.csr_preprocess_hybrid(torch.from_numpy(indptr), torch.from_numpy(indices), 777, tau=60000)
    assert empty.plan.num_ksteps == 0
    # an empty plan leaves the whole matrix in the residual: its tensors ARE the reference handle, and exact-fp32 mode works
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    plain = voltrix.csr_preprocess(torch.from_numpy(indptr), torch.from_numpy(indices), 777)
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(empty.residual, plain))
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact")
    empty.hash_tag = "degenerate_exact"
    _assert_close(voltrix.spmm_two_level(empty, feat32.cuda()), indptr, indices, feat32, 777, "exact")
    monkeypatch.delenv("VOLTRIX_FP32_MODE")
    # below VOLTRIX_HYBRID_MIN_SHARE csr_preprocess attaches no side-car
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "1.5")   # out of reach
    dropped = voltrix.csr_preprocess(torch.from_numpy(indptr), torch.from_numpy(indices), 777)
    assert voltrix.sidecar.lookup(dropped[1]) == (True, None)
    # universe above 2^22 columns: the plan is empty by design, everything stays in the window format
    wide_cols = hybrid.MAX_PLAN_COLS + 1000
    w_indptr, w_indices = _random_csr(200, 30, seed=12, ncols=wide_cols)
    ri, rx, plan = hybrid.build_panel_plan(torch.from_numpy(w_indptr).cuda(), torch.from_numpy(w_indices).cuda(), 200, wide_cols)
    assert plan.num_ksteps == 0 and torch.equal(rx.cpu(), torch.from_numpy(w_indices)) and plan.num_resid_edges == len(w_indices)
    # ids outside the declared universe (a rectangular operand declared too narrow): empty plan, like csr_preprocess's retry
    ri, rx, plan = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), 777, 100, 4, 2, 2)
    assert plan.num_ksteps == 0 and torch.equal(rx.cpu(), torch.from_numpy(indices))


def test_side_car_follows_copies_only_when_told_and_survives_save_load(cuda_device, tmp_path, monkeypatch):
    """VERDICT r3 item 6.  The side-car is recorded for the MEMORY of ``hspa_packed`` (views keep it); a clone / a reloaded
    handle has none until ``copy_side_car`` / ``load_handle`` re-attach it; a big handle without any record makes
    ``voltrix.spmm`` warn (once) instead of silently running the slower format; all forms give the same bits."""
    import warnings

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = "sidecar_roundtrip"
    two = voltrix.two_level_of(handle[1])
    assert two is not None and two.plan.num_ksteps > 0
    feat = torch.randint(-3, 4, (n, 64), device="cuda").half()       # integers: every summation order gives the same bits
    want = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    with voltrix.utils.KernelTimer() as timer:
        voltrix.spmm(handle[0], handle[1].view(-1), handle[2], num_nodes=n, num_edges=e, feat=feat)   # a view: still two-level
    assert any("spmm_panel" in k for k in timer.summary())
    clone = tuple(t.clone() for t in handle)
    clone[1].hash_tag = "sidecar_roundtrip_clone"
    with voltrix.utils.KernelTimer() as timer:
        got = voltrix.spmm(*clone, num_nodes=n, num_edges=e, feat=feat)
    assert not any("spmm_panel" in k for k in timer.summary()) and torch.equal(got, want)        # window format, same product
    assert voltrix.copy_side_car(handle[1], clone[1]) and voltrix.two_level_of(clone[1]) is two
    with voltrix.utils.KernelTimer() as timer:
        got = voltrix.spmm(*clone, num_nodes=n, num_edges=e, feat=feat)
    assert any("spmm_panel" in k for k in timer.summary()) and torch.equal(got, want)
    # save -> load: the reference tensors byte for byte, the side-car rebuilt from the file (nothing is recomputed)
    path = str(tmp_path / "handle.pt")
    voltrix.save_handle(path, handle, n)
    loaded = voltrix.load_handle(path)
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(loaded, handle))
    two_l = voltrix.two_level_of(loaded[1])
    assert two_l is not None and two_l is not two and two_l.plan.num_ksteps == two.plan.num_ksteps
    assert torch.equal(two_l.plan.panel_bits.view(torch.int32), two.plan.panel_bits.view(torch.int32))
    assert loaded[1].hash_tag == "sidecar_roundtrip"
    with voltrix.utils.KernelTimer() as timer:
        got = voltrix.spmm(*loaded, num_nodes=n, num_edges=e, feat=feat)
    assert any("spmm_panel" in k for k in timer.summary()) and torch.equal(got, want)
    # a handle saved with the decision "window format" comes back with that decision, not as an unknown
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    plain = voltrix.csr_preprocess(indptr, indices, n)
    voltrix.save_handle(path, plain, n)
    assert voltrix.sidecar.lookup(voltrix.load_handle(path)[1]) == (True, None)
    # the warning: auto mode, no record, a graph of side-car size (faked through the edge count argument)
    monkeypatch.setenv("VOLTRIX_HYBRID", "auto")
    voltrix.sidecar._WARNED[0] = False
    unknown = tuple(t.clone() for t in handle)
    unknown[1].hash_tag = "sidecar_roundtrip_unknown"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        voltrix.spmm(*unknown, num_nodes=n, num_edges=e, feat=feat)                               # small graph: silent
        assert not [w for w in caught if "copy_side_car" in str(w.message)]
        monkeypatch.setattr(voltrix.hybrid, "AUTO_MIN_EDGES", 1)
        monkeypatch.setattr(voltrix.hybrid, "AUTO_MIN_ROWS", 1)
        monkeypatch.setattr(voltrix.hybrid, "AUTO_MIN_MEAN_DEGREE", 0)
        for _ in range(2):
            voltrix.spmm(*unknown, num_nodes=n, num_edges=e, feat=feat)
    assert len([w for w in caught if "copy_side_car" in str(w.message)]) == 1
    voltrix.sidecar._WARNED[0] = False


def test_xcd_ranges_of_equal_work_give_the_same_bits(cuda_device, monkeypatch):
    """Round 4: both kernels of the two-level step take their XCD ranges from the work (hybrid.balance_xcd_ranges) instead
    of NP / 8 panels each.  A block-model graph (communities of very different sizes: the busiest eighth of the rows holds
    well above 1/8 of the work): the ranges differ from the equal-count ones, their work is balanced, and the result --
    window pair and one-launch form -- has the same bits as with equal-count ranges (the order of a row's sums is unchanged:
    ranges only decide WHICH CU runs a panel)."""
    import numpy as np

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_FUSED", "1")
    indptr, indices, _ = synth_graphs.generate("reddit_sbm", device="cuda", scale=0.25)
    n = indptr.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr.cpu(), indices.cpu(), n)
    two.hash_tag = "xcd_balanced"
    plan = two.plan
    assert plan.xcd_ptr is not None and two.window_xcd_ptr is not None and two.fused is not None
    xp = plan.xcd_ptr.cpu().numpy()
    assert xp[0] == 0 and xp[8] == plan.num_panels and (np.diff(xp) >= 0).all() and plan.max_panels_per_xcd == np.diff(xp).max()
    equal = np.minimum(np.arange(9) * ((plan.num_panels + 7) // 8), plan.num_panels)
    assert (xp != equal).any()
    ksteps = np.diff(plan.panel_ptr.cpu().numpy())
    nst = (np.diff(two.blk_offsets.cpu().numpy().astype(np.int64)) + 3) // 4
    nst = np.concatenate([nst, np.zeros(plan.num_panels * 32 - nst.size, np.int64)]).reshape(plan.num_panels, 32).sum(1)
    work = hybrid.KSTEP_COST_X10 / 10 * ksteps + nst

    def busiest(ptr):
        per = np.array([work[ptr[x]:ptr[x + 1]].sum() for x in range(8)])
        return per.max() / per.mean()

    assert busiest(xp) < min(1.12, busiest(equal) - 0.05), (busiest(xp), busiest(equal))   # 114 panels: 14 per XCD
    assert (two.window_xcd_ptr.cpu().numpy() == np.minimum(xp.astype(np.int64) * 32, (n + 15) // 16)).all()
    order = plan.panel_order.cpu().numpy()
    for x in range(8):
        assert sorted(order[xp[x]:xp[x + 1]]) == list(range(xp[x], xp[x + 1]))

    feat = torch.randn(n, 128, device=cuda_device).half()
    balanced = voltrix.spmm_two_level(two, feat).clone()
    monkeypatch.setenv("VOLTRIX_FUSED", "0")
    balanced_pair = voltrix.spmm_two_level(two, feat).clone()
    saved = (plan.xcd_ptr, plan.max_panels_per_xcd, plan.panel_order, two.window_xcd_ptr)
    plan.xcd_ptr, plan.max_panels_per_xcd, two.window_xcd_ptr = None, 0, None
    plan.panel_order = hybrid.longest_first_order(plan.panel_ptr)
    two.hash_tag = "xcd_equal_counts"
    assert torch.equal(voltrix.spmm_two_level(two, feat), balanced_pair)
    monkeypatch.setenv("VOLTRIX_FUSED", "1")
    assert torch.equal(voltrix.spmm_two_level(two, feat), balanced)
    plan.xcd_ptr, plan.max_panels_per_xcd, plan.panel_order, two.window_xcd_ptr = saved
    ref = torch.sparse.mm(torch.sparse_csr_tensor(indptr.long(), indices.long(), torch.ones(indices.numel(), device="cuda"),
                                                  size=(n, n)), feat.float())
    assert float((balanced - ref).norm() / ref.norm()) < 1e-3 and float((balanced_pair - ref).norm() / ref.norm()) < 1e-3


@pytest.mark.parametrize("waves,rb,tile,feat_dim", [(8, 4, (128, 3, 1), 128), (8, 4, (128, 3, 1), 264), (4, 4, (64, 4, 1), 64),
                                                    (8, 4, (64, 4, 2), 40), (8, 2, (128, 4, 1), 128),
                                                    (8, 2, (128, 4, 17), 128), (8, 4, (128, 3, 17), 264)])
@pytest.mark.parametrize("cap", [1, 3, 1000])
def test_panel_kernel_in_pieces(cuda_device, waves, rb, tile, feat_dim, cap):
    """Round 4: the panel kernel over PIECES of panels (PanelArgs::parts) -- cut panels leave partial tiles that
    combine_panel_partials adds in slot order.  Small integers: the same bits as one workgroup per panel in all three output
    modes (every partial sum is an integer below 2^24, so the order of the pieces cannot show).  Random data: within the
    stated bound of the oracle, and the same bits run to run.  cap = 1: every k-step its own piece; 1000: nothing is cut
    (table without slots); several column slabs (F = 264), two k-steps per ring slot, the N % panel tail."""
    n = 1100
    indptr, indices = _random_csr(n, 90, seed=waves + rb + feat_dim)
    _, _, plan = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), n, None, waves,
                                         rb, 2)
    shared = oracle_np.panel_to_edges(plan.panel_ptr.cpu().numpy(), plan.panel_cols.cpu().numpy(),
                                      plan.panel_bits.view(torch.int32).cpu().numpy().view(np.uint32), n, waves, rb)
    s_indptr, s_indices = _edges_to_csr(shared, n)
    torch.manual_seed(feat_dim + cap)
    ints = torch.randint(-3, 4, (n, feat_dim), device=cuda_device).half()
    prior = torch.randint(-50, 50, (n, feat_dim), device=cuda_device).float()

    def run(feat, accumulate):
        out = prior.clone() if accumulate else torch.full((n, feat_dim), float("nan"), device=cuda_device)
        hybrid.launch_panel(plan, feat, out, accumulate=accumulate, tile=tile)
        return out

    whole = {acc: run(ints, acc) for acc in (0, 1, 2)}
    plan.parts = hybrid.panel_parts(plan.panel_ptr, cap)
    assert (plan.parts.num_slots > 0) == (cap < 1000) and plan.parts.num_parts >= plan.num_panels
    for acc in (0, 1, 2):
        assert torch.equal(run(ints, acc), whole[acc]), acc
    feat = torch.randn(n, feat_dim).half()
    out = run(feat.cuda(), 0)
    _assert_close(out, s_indptr, s_indices, feat.float(), n, "fp16")
    assert torch.equal(run(feat.cuda(), 0), out)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    out_side = torch.full((n, feat_dim), float("nan"), device=cuda_device)
    hybrid.launch_panel(plan, feat.cuda(), out_side, accumulate=0, tile=tile, stream=side.cuda_stream)   # combine on that stream too
    side.synchronize()
    assert torch.equal(out_side, out)


def test_two_level_operator_with_pieces(cuda_device, monkeypatch, tmp_path):
    """The operator with a part table on the plan (forced: the test graph's panels are short): window pair + atomic join +
    both combine passes, one-stream form, graph capture, save / load.  Integers: bit-equal to the table-free plan."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("reddit_sbm", device="cuda", scale=0.1)
    n = indptr.numel() - 1
    two = voltrix.csr_preprocess_hybrid(indptr.cpu(), indices.cpu(), n)
    two.hash_tag = "pieces"
    plan = two.plan
    assert plan.num_ksteps > 0
    ints = torch.randint(-3, 4, (n, 128), device=cuda_device).half()
    plan.parts = None
    ref = voltrix.spmm_two_level(two, ints).clone()
    ref_one_stream = voltrix.spmm_two_level(two, ints, concurrent=False).clone()
    assert torch.equal(ref, ref_one_stream)
    cap = max(1, int(torch.diff(plan.panel_ptr).max()) // 3)
    plan.parts = hybrid.panel_parts(plan.panel_ptr, cap, plan.xcd_ptr)
    assert plan.parts.num_cuts > 0
    assert torch.equal(voltrix.spmm_two_level(two, ints), ref)
    assert torch.equal(voltrix.spmm_two_level(two, ints, concurrent=False), ref)
    feat = torch.randn(n, 128, device=cuda_device).half()
    out = voltrix.spmm_two_level(two, feat).clone()
    _assert_close(out, indptr.cpu().numpy(), indices.cpu().numpy(), feat.float().cpu(), n, "fp16")
    assert torch.equal(voltrix.spmm_two_level(two, feat), out)            # run to run
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    static_in = feat.clone()
    with torch.cuda.graph(graph):
        static_out = voltrix.spmm_two_level(two, static_in)
    static_in.copy_(ints)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref)
    path = str(tmp_path / "handle.pt")          # the part table travels as its bound and is rebuilt on load
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    handle = voltrix.csr_preprocess(indptr.cpu(), indices.cpu(), n)
    side = voltrix.two_level_of(handle[1])
    side.plan.parts = hybrid.panel_parts(side.plan.panel_ptr, cap, side.plan.xcd_ptr)
    voltrix.save_handle(path, handle, n)
    loaded = voltrix.load_handle(path)
    lp = voltrix.two_level_of(loaded[1]).plan.parts
    assert lp is not None and lp.cap == cap and torch.equal(lp.parts, side.plan.parts.parts) and torch.equal(lp.cuts, side.plan.parts.cuts)
    loaded[1].hash_tag = "pieces_loaded"
    assert torch.equal(voltrix.spmm(*loaded, num_nodes=n, num_edges=indices.numel(), feat=ints), ref)


def test_slim_handle_runs_on_the_side_car_alone_and_refuses_the_window_paths(cuda_device, monkeypatch):
    """voltrix.slim_handle: after the side-car has been built the reference tensors of the whole matrix can be dropped -- the
    operator gives the same bits from the 4-element stand-ins (reddit-like headline: 617 MB freed), and refuses the paths
    that would need the dropped TC blocks instead of reading four elements as a handle."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    handle[1].hash_tag = "slim_test"
    assert voltrix.two_level_of(handle[1]) is not None
    feat = torch.randn(n, 128, device=cuda_device).half()
    ref = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat).clone()
    before = sum(t.numel() * t.element_size() for t in handle)
    slim = voltrix.slim_handle(handle)
    assert slim[0] is handle[0] and slim[1].numel() == 4 and slim[2].numel() == 4
    assert sum(t.numel() * t.element_size() for t in slim) < before / 10
    del handle
    assert torch.equal(voltrix.spmm(*slim, num_nodes=n, num_edges=e, feat=feat), ref)
    _assert_close(ref, indptr.numpy(), indices.numpy(), feat.float().cpu(), n, "fp16")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    with pytest.raises(AssertionError, match="slimmed"):
        voltrix.spmm(*slim, num_nodes=n, num_edges=e, feat=feat)
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact")
    with pytest.raises(AssertionError, match="slimmed"):
        voltrix.spmm(*slim, num_nodes=n, num_edges=e, feat=feat.float())
    monkeypatch.delenv("VOLTRIX_FP32_MODE")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")                         # a handle without a side-car is returned unchanged
    no_side_car = voltrix.csr_preprocess(indptr, indices, n)
    assert voltrix.slim_handle(no_side_car)[1] is no_side_car[1]


def test_pipelined_loop_gives_the_classic_loop_bits(cuda_device):
    """Round 5: PanelTile<..., PIPE> walks the k-steps in the same order per accumulator -- the same bits as the classic loop on
    random data, for panels shorter than, equal to and longer than the ring (0 .. 40 k-steps), all three output modes."""
    assert hybrid.KSTEPS_PIPELINED == 17
    n = 2300
    indptr, indices = _random_csr(n, 70, seed=5)
    indptr[700:1300] = indptr[700]      # panels without k-steps
    for waves, rb, classic, piped in ((8, 4, (128, 3, 1), (128, 3, 17)), (8, 4, (128, 4, 1), (128, 4, 17)),
                                      (8, 2, (128, 4, 1), (128, 4, 17)), (8, 2, (128, 6, 1), (128, 6, 17)),
                                      (8, 4, (64, 4, 2), (64, 4, 17))):
        for tau in (1, 2, 9):
            _, _, plan = hybrid.build_panel_plan(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), n, None,
                                                 waves, rb, tau)
            feat_dim = 200 if classic[0] == 128 else 64
            prior = torch.randn(n, feat_dim, device=cuda_device)
            for dtype in (torch.float16, torch.bfloat16):
                feat = torch.randn(n, feat_dim, device=cuda_device).to(dtype)
                for acc in (0, 1, 2):
                    a, b = prior.clone(), prior.clone()
                    hybrid.launch_panel(plan, feat, a, accumulate=acc, tile=classic)
                    hybrid.launch_panel(plan, feat, b, accumulate=acc, tile=piped)
                    assert torch.equal(a, b), (waves, rb, tau, acc, dtype)


def test_panel_dominated_graphs_get_256_row_panels(cuda_device, monkeypatch):
    """Round 5: when the panel kernel of the default 8 x 4 plan would be the step's critical path (hybrid.panel_dominated),
    ``csr_preprocess`` builds 8 x 2 = 256-row panels instead and the operator runs them with the pipelined tile; exact on
    integers against hipSPARSE.  The one-launch form (VOLTRIX_FUSED) keeps its 8 x 4 panels."""
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    indptr, indices, _ = synth_graphs.generate("protein_like", device="cuda", scale=0.1)
    n, nnz = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    two = voltrix.two_level_of(handle[1])
    assert two is not None and (two.plan.waves, two.plan.row_blocks, two.plan.panel_rows) == (8, 2, 256)
    assert hybrid.default_panel_tile(128, 8, 2) == (128, 4, hybrid.KSTEPS_PIPELINED)
    for f in (32, 128, 264):
        feat = torch.randint(-3, 4, (n, f), device="cuda").half()
        ref = torch.sparse_csr_tensor(indptr, indices, torch.ones(nnz, device="cuda"), size=(n, n)) @ feat.float()
        assert torch.equal(voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat), ref), f
    plan84 = hybrid.PanelPlan(panel_ptr=None, panel_cols=None, panel_bits=None, panel_order=None, num_nodes=n, waves=8,
                              row_blocks=4, tau=3, num_ksteps=1000, num_shared_edges=0, num_resid_edges=32 * 6600)
    assert not hybrid.panel_dominated(plan84)                  # 6.6 x 1000 against 2 x 6600 stages
    plan84.num_resid_edges = 32 * 3300
    assert hybrid.panel_dominated(plan84)
    monkeypatch.setenv("VOLTRIX_FUSED", "1")
    fused = voltrix.two_level_of(voltrix.csr_preprocess_device(indptr, indices, n)[1])
    assert (fused.plan.waves, fused.plan.row_blocks) == (8, 4)
