"""GPU: every BASELINE.json configuration at its STATED size, through the C-ABI.

No CPU oracle finishes at these sizes, so the checks are the size-independent properties of ``C = A @ B`` for a binary A:

  (1) ``A @ 1 == degree`` exactly (sums of 1.0 up to 4e5 are exact in fp32) -- every row, so a 32-bit overflow of
      ``row * F * 4`` (products-like F=512: N*F*4 = 5.0e9 > 2^31; the reference overflows at spmm_kernels.cuh:1568,1688),
      of ``8 * block`` (power-law: 2e8 TC blocks, hind is 6.4 GB) or of a row of B (papers-like: B is 28 GB) shows up;
  (2) linearity, bit-exact on small-integer operands (every partial sum is an integer below 2^24);
  (3) checksum of checksums: the column sums of ``A @ x`` equal ``(A^T 1) . x`` computed from the in-degree of every column.

Memory is bounded by checking in row chunks (papers-like: B 28 GB, C 57 GB per result).

Round 4 -- the STATED floating-point tolerance at the stated sizes (``test_*_stated_tolerance``): a random fp16 operand, and
the reference's own oracle call (``torch.sparse.mm(csr(ones), feat)`` on the CPU, oracle/torch_ref.py) evaluated for a SAMPLE
of the rows -- 2,048 random ones plus the 64 of highest degree (21 k on reddit-like, 4e5 on the power-law graph) -- on their
sub-CSR with the referenced rows of B compacted.  Bars, as in tests/test_gpu_spmm.py: element-wise
``|out - ref| <= (2^-11 + deg 2^-23) (A |B|) + deg 2^-25`` and, the operand being fp16 already (only the accumulation order
differs), ``<= deg 2^-23 (A |B|)``; norm-wise ``||out - ref|| / ||ref|| <= 1e-3`` over the sampled rows.
"""
import functools

import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from oracle import torch_ref
from voltrix import capi

pytestmark = pytest.mark.gpu

CHUNK = 1 << 21


@functools.lru_cache(maxsize=None)
def _graph(workload):
    """One generation per workload and module (reddit 0.5 GB, products 0.5 GB, power-law 6.4 GB, papers 6.5 GB of CSR)."""
    indptr, indices, _ = synth_graphs.generate(workload, device="cuda")
    return indptr, indices


def teardown_module(module):
    _graph.cache_clear()
    torch.cuda.empty_cache()


def _launch(handle, n, e, feat, out, tile, order=0):
    rc = capi.launch_spmm(handle[0].data_ptr(), handle[1].data_ptr(), handle[2].data_ptr(), n, e, feat.shape[1],
                          feat.data_ptr(), out.data_ptr(), True, tile, torch.cuda.current_stream().cuda_stream, order)
    assert rc == 0, f"voltrix_launch_spmm_f16_tile rc={rc}"
    return out


def _rows_equal(out, per_row):
    """out[i, :] == per_row[i] for every i (chunked: no N x F temporary)."""
    for r in range(0, out.shape[0], CHUNK):
        if not bool((out[r:r + CHUNK] == per_row[r:r + CHUNK, None]).all()):
            return False
    return True


def _equal_sum(a, b, c):
    """a == b + c element-wise (chunked)."""
    for r in range(0, a.shape[0], CHUNK):
        if not torch.equal(a[r:r + CHUNK], b[r:r + CHUNK] + c[r:r + CHUNK]):
            return False
    return True


def _column_sums(t, weights=None):
    """float64 column sums of t (optionally row-weighted), chunked."""
    acc = torch.zeros(t.shape[1], dtype=torch.float64, device=t.device)
    for r in range(0, t.shape[0], CHUNK):
        blk = t[r:r + CHUNK].double()
        if weights is not None:
            blk *= weights[r:r + CHUNK, None]
        acc += blk.sum(dim=0)
    return acc


def _small_ints(n, f, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    return torch.empty(n, f, dtype=torch.float16, device="cuda").random_(-3, 4, generator=gen)


def _check_properties(run, indptr, indices, n, f, linearity=True):
    """``run(feat) -> out`` (a fresh or reused fp32 [n, f] tensor)."""
    deg = (indptr[1:] - indptr[:-1]).float()
    assert int(deg.max()) < 2 ** 24
    ones = torch.ones(n, f, dtype=torch.float16, device="cuda")
    out = run(ones)
    assert _rows_equal(out, deg), "A @ 1 != degree"
    del ones, out
    x = _small_ints(n, f, 1)
    ox = run(x).clone()
    col_deg = torch.bincount(indices.long(), minlength=x.shape[0]).double()
    assert torch.equal(_column_sums(ox), _column_sums(x, col_deg)), "column checksum"
    del col_deg
    if linearity:
        y = _small_ints(n, f, 2)
        oy = run(y).clone()
        y += x                                   # exact: |x + y| <= 6
        del x
        oxy = run(y)
        assert _equal_sum(oxy, ox, oy), "A @ (x + y) != A @ x + A @ y"


def _window_runner(handle, n, e, f, tile, order_chunk=512):
    out = torch.empty(n, f, dtype=torch.float32, device="cuda")
    order = torch.empty((n + 15) // 16, dtype=torch.int32, device="cuda")
    capi.launch_window_order(handle[0], n, order, torch.cuda.current_stream().cuda_stream, order_chunk)

    def run(feat):
        out.fill_(float("nan"))                  # every element must be written by the kernel
        return _launch(handle, n, e, feat, out, tile, order.data_ptr())

    return run


def test_products_like_f512_full_size(cuda_device):
    """BASELINE config 3: N = 2,449,029, 123.7 M edges, F = 512 fp16.  C is 5.0 GB: row * F * 4 passes 2^31 at row 1.05 M."""
    indptr, indices = _graph("products_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 512
    assert n == 2449029 and e == int(synth_graphs.target_degrees("products_like", device="cuda").sum())
    assert n * f * 4 > 2 ** 31
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    _check_properties(_window_runner(handle, n, e, f, capi.default_tile(f, True)), indptr, indices, n, f)


def test_powerlaw_4m_f256_full_size(cuda_device):
    """BASELINE config 4: 4 M rows, density 1e-4 (1.6e9 edges), Zipf alpha = 2 degrees up to 4e5, F = 256 fp16.
    Windows reach tens of thousands of TC blocks; hind alone is 6.4 GB (8 * block passes 2^31 bytes early)."""
    indptr, indices = _graph("powerlaw_4m")
    n, e, f = indptr.numel() - 1, indices.numel(), 256
    assert n == 4000000 and abs(e - 1.6e9) < 2e6
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    blocks = handle[0][1:] - handle[0][:-1]
    assert int(blocks.max()) > 20000 and int(handle[0][-1]) * 32 > 2 ** 32
    _check_properties(_window_runner(handle, n, e, f, capi.default_tile(f, True), order_chunk=2048), indptr, indices, n, f)


def test_papers_like_f128_one_gpu(cuda_device):
    """BASELINE config 5 on ONE GPU: N = 111,059,956, 1.6e9 edges, F = 128 fp16 (B = 28.4 GB, C = 56.9 GB)."""
    indptr, indices = _graph("papers_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    assert n == 111059956 and abs(e - 1615685872) < 2e6
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    assert n * f * 2 > 2 ** 34
    # linearity needs three 57 GB results at once: covered by the other configs; here A @ 1 and the column checksum
    _check_properties(_window_runner(handle, n, e, f, capi.default_tile(f, True)), indptr, indices, n, f, linearity=False)


def test_two_level_reddit_like_full_size(cuda_device, monkeypatch):
    """BASELINE config 2 (the headline) at full size in the two-level format, through the operator (``voltrix.spmm``):
    same properties, and bit-equality with the window format on integer operands."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices = _graph("reddit_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    assert n == 232965 and abs(e - 114615892) < 100000
    indptr_c, indices_c = indptr.cpu(), indices.cpu()
    two = voltrix.csr_preprocess_hybrid(indptr_c, indices_c, n)
    assert two.plan.panel_rows == 512 and two.plan.num_shared_edges > 0.4 * e
    assert two.plan.num_shared_edges + two.plan.num_resid_edges == e

    def run(feat):
        return voltrix.spmm_two_level(two, feat)

    _check_properties(run, indptr, indices, n, f)
    win = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    x = _small_ints(n, f, 7)
    a = run(x)
    b = _window_runner(win, n, e, f, capi.default_tile(f, True))(x)
    assert torch.equal(a, b)
    # round 4: the schedules of the step at full size -- XCD ranges of equal work came with the handle; panels in pieces of at
    # most 100 k-steps (the product's bound, S / 256 = 713, cuts none of this graph's panels): 2,000-odd pieces, every panel
    # cut, partial tiles summed in slot order -- the same bits on integers, and the size-independent properties again
    from voltrix import hybrid

    plan = two.plan
    assert plan.xcd_ptr is not None and two.window_xcd_ptr is not None and plan.parts is None
    plan.parts = hybrid.panel_parts(plan.panel_ptr, 100, plan.xcd_ptr)
    assert plan.parts.num_cuts > 400 and plan.parts.num_parts > 1800
    assert int(plan.parts.parts[:, 2].sum()) == plan.num_ksteps and int(plan.parts.parts[:, 2].max()) <= 100
    assert torch.equal(run(x), a)
    _check_properties(run, indptr, indices, n, f)
    plan.parts = None


def test_format_policy_is_decided_in_csr_preprocess_at_full_size(cuda_device, monkeypatch):
    """The default mode (VOLTRIX_HYBRID=auto) at BASELINE size: ``csr_preprocess_device`` attaches the two-level side-car to the
    reddit-like graph (55 % of the edges in shared columns) and not to its uniform-column variant (29 %: the plan builder stops
    after its count phase -- nothing of the losing form is built); the operator then runs the chosen form without any timing
    of its own, the one-launch form of the same product gives the same bits on integers, and the opt-in ``tune`` mode still
    times both forms on its first call."""
    from voltrix import hybrid

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.delenv("VOLTRIX_HYBRID", raising=False)
    monkeypatch.delenv("VOLTRIX_HYBRID_MIN_SHARE", raising=False)
    indptr, indices, _ = synth_graphs.generate("reddit_uniform", device="cuda")
    n, e = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    assert voltrix.two_level_of(handle[1]) is None
    r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, min_share=hybrid.min_shared_fraction())
    assert plan.num_ksteps == 0 and r_indptr is indptr and r_indices is indices     # stopped after the count phase
    assert 0.2 < plan.num_shared_edges / e < 0.4
    del handle, indptr, indices, r_indptr, r_indices

    indptr, indices = _graph("reddit_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = "policy_test"
    two = voltrix.two_level_of(handle[1])
    assert two is not None and two.plan.num_shared_edges > 0.5 * e and two.fused is None and two.format_choice == {}
    assert hybrid.two_level_bytes(two) < 1.2 * hybrid.handle_bytes(handle)
    x = _small_ints(n, f, 11)
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)
    assert two.format_choice == {}                                                     # no timed comparison happened
    two.fused = hybrid.build_fused_records(two.blk_offsets, two.hspa_packed, two.hind, n)
    one = torch.empty_like(out)
    hybrid.launch_fused(two.plan, two.fused, x, one)
    assert torch.equal(out, one)
    two.fused = None
    monkeypatch.setenv("VOLTRIX_HYBRID", "tune")
    monkeypatch.setenv("VOLTRIX_TUNED_STORE", "/tmp/voltrix_policy_test_tuned.json")
    out_t = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)
    assert list(two.format_choice.values())[0] in ("two-level", "window") and torch.equal(out_t, out)


# ---- the stated floating-point tolerance at the stated sizes (round 4) ------------------------------------------------------

def _random_fp16(rows, f, seed):
    """randn fp16 [rows, f], generated in chunks (no fp32 copy of a 28 GB operand)."""
    gen = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty(rows, f, dtype=torch.float16, device="cuda")
    for r in range(0, rows, CHUNK):
        out[r:r + CHUNK] = torch.randn(min(CHUNK, rows - r), f, generator=gen, device="cuda").half()
    return out


def _sample_rows(indptr, n, seed, num_random=2048, num_top=64):
    deg = indptr[1:] - indptr[:-1]
    gen = torch.Generator(device="cuda").manual_seed(seed)
    rnd = torch.randint(0, n, (num_random,), generator=gen, device="cuda")
    top = torch.topk(deg, num_top).indices
    return torch.unique(torch.cat([rnd, top, torch.tensor([0, n - 1], device="cuda")]))     # sorted; first and last row too


def _sub_problem(indptr, indices, rows, feat):
    """Sub-CSR of ``rows`` with the referenced rows of ``feat`` compacted: (indptr int32 cpu, indices int32 cpu, B fp32 cpu)."""
    ip = indptr.long()
    start, cnt = ip[rows], ip[rows + 1] - ip[rows]
    sub_ptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device="cuda")
    sub_ptr[1:] = torch.cumsum(cnt, 0)
    total = int(sub_ptr[-1])
    pos = torch.arange(total, device="cuda") - torch.repeat_interleave(sub_ptr[:-1], cnt) + torch.repeat_interleave(start, cnt)
    cols = indices[pos].long()
    used = torch.unique(cols)
    return (sub_ptr.to(torch.int32).cpu(), torch.searchsorted(used, cols).to(torch.int32).cpu(), feat[used].float().cpu())


def _assert_stated_tolerance(out_rows, sub, what):
    sub_ptr, sub_idx, b_sub = sub
    rows = sub_ptr.numel() - 1
    ref = torch_ref.spmm(sub_ptr, sub_idx, b_sub, rows).double().numpy()          # the reference's oracle call, CPU fp32
    aabs = (torch.sparse_csr_tensor(sub_ptr, sub_idx, torch.ones(sub_idx.numel(), dtype=torch.float64),
                                    size=(rows, b_sub.shape[0])) @ b_sub.abs().double()).numpy()
    deg = np.diff(sub_ptr.numpy().astype(np.int64)).astype(np.float64)[:, None]
    out = out_rows.double().cpu().numpy()
    assert not np.isnan(out).any(), what
    err = np.abs(out - ref)
    assert (err <= (2.0 ** -11 + deg * 2.0 ** -23) * aabs + deg * 2.0 ** -25 + 1e-30).all(), what      # the stated bound
    assert (err <= deg * 2.0 ** -23 * aabs + 1e-30).all(), what                # fp16 operand: accumulation order only
    rel = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    assert rel <= 1e-3, (what, rel)
    return float(rel), int(deg.max())


def _stated_tolerance_window(workload, f, seed, n_expected):
    indptr, indices = _graph(workload)
    n, e = indptr.numel() - 1, indices.numel()
    assert n == n_expected
    feat = _random_fp16(n, f, seed)
    rows = _sample_rows(indptr, n, seed)
    sub = _sub_problem(indptr, indices, rows, feat)
    handle = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    out = _window_runner(handle, n, e, f, capi.default_tile(f, True))(feat)
    rel, dmax = _assert_stated_tolerance(out[rows], sub, f"{workload} window format")
    return rel, dmax


def test_reddit_like_f128_stated_tolerance(cuda_device, monkeypatch):
    """BASELINE config 2 (headline), random fp16 operand, rows of degree up to 21 k: the window format through the C-ABI, the
    operator's default (two-level side-car, panel + window kernels joined by float atomics) and the one-launch form."""
    from voltrix import hybrid

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.delenv("VOLTRIX_HYBRID", raising=False)
    monkeypatch.delenv("VOLTRIX_FUSED", raising=False)
    rel, dmax = _stated_tolerance_window("reddit_like", 128, 21, 232965)
    assert dmax > 15000
    indptr, indices = _graph("reddit_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    feat = _random_fp16(n, f, 21)
    rows = _sample_rows(indptr, n, 21)
    sub = _sub_problem(indptr, indices, rows, feat)
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = "stated_tolerance_reddit"
    two = voltrix.two_level_of(handle[1])
    assert two is not None                                                            # the default IS the two-level form here
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    _assert_stated_tolerance(out[rows], sub, "reddit_like two-level default")
    two.fused = hybrid.build_fused_records(two.blk_offsets, two.hspa_packed, two.hind, n)
    monkeypatch.setenv("VOLTRIX_FUSED", "1")
    one = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    _assert_stated_tolerance(one[rows], sub, "reddit_like one-launch form")


def test_products_like_f512_stated_tolerance(cuda_device):
    """BASELINE config 3: N = 2,449,029, F = 512 (C is 5 GB: sampled rows on both sides of the 2^31-byte offset)."""
    rel, dmax = _stated_tolerance_window("products_like", 512, 22, 2449029)
    assert dmax > 5000


def test_powerlaw_4m_f256_stated_tolerance(cuda_device):
    """BASELINE config 4: rows of degree up to 4e5 (a window of > 20 k TC blocks), F = 256."""
    rel, dmax = _stated_tolerance_window("powerlaw_4m", 256, 23, 4000000)
    assert dmax > 300000


def test_papers_like_f128_stated_tolerance(cuda_device):
    """BASELINE config 5 on one GPU: B = 28 GB, C = 57 GB; the sampled rows reference rows of B all over it."""
    _stated_tolerance_window("papers_like", 128, 24, 111059956)


@pytest.mark.parametrize("path", ["separable", "value_plane"])
def test_weighted_reddit_like_f128_stated_tolerance(cuda_device, path, monkeypatch):
    """Round 6: the same values through BOTH weighted paths -- detected as r_i c_j (the binary operator with its two-level side-car
    between two row scalings; the default) and forced through the general value plane.
    The weighted product (voltrix/weighted.py; no reference counterpart) at the headline size: values = the symmetric-
    normalised adjacency, random fp16 B, sampled rows (degree up to 21 k) against torch.sparse.mm with fp32 values on the CPU.
    A and B are both rounded to fp16: |out - ref| <= (2^-10 + deg 2^-23) (|A| |B|) + deg 2^-25 max|a|, norm-wise <= 1e-3."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices = _graph("reddit_like")
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    deg = (indptr[1:] - indptr[:-1]).long()
    d_out = deg.float().clamp(min=1)
    d_in = torch.bincount(indices.long(), minlength=n).float().clamp(min=1)
    values = torch.repeat_interleave(d_out.rsqrt(), deg) * d_in.rsqrt()[indices.long()]
    handle = voltrix.csr_preprocess_weighted(indptr, indices, values, n, separable="auto" if path == "separable" else False)
    assert handle.separable == (path == "separable")
    if path == "separable":
        assert voltrix.two_level_of(handle.hspa_packed) is not None, "the binary operator of this graph carries the two-level side-car"
    feat = _random_fp16(n, f, 31)
    out = voltrix.spmm_weighted(handle, feat, hash_tag=f"stated_tolerance_weighted_reddit_{path}")
    rows = _sample_rows(indptr, n, 31)
    sub_ptr, sub_idx, b_sub = _sub_problem(indptr, indices, rows, feat)
    ip = indptr.long()
    cnt = ip[rows + 1] - ip[rows]
    pos = (torch.arange(int(cnt.sum()), device="cuda") - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
           + torch.repeat_interleave(ip[rows], cnt))
    v_sub = values[pos].cpu()
    k = rows.numel()
    a32 = torch.sparse_csr_tensor(sub_ptr, sub_idx, v_sub, size=(k, b_sub.shape[0]))
    ref = (a32 @ b_sub).double().numpy()                                   # torch.sparse.mm with values: the oracle
    aabs = (torch.sparse_csr_tensor(sub_ptr, sub_idx, v_sub.abs().double(), size=(k, b_sub.shape[0])) @ b_sub.abs().double()).numpy()
    dg = np.diff(sub_ptr.numpy().astype(np.int64)).astype(np.float64)[:, None]
    got = out[rows].double().cpu().numpy()
    err = np.abs(got - ref)
    assert not np.isnan(got).any() and int(dg.max()) > 15000
    assert (err <= (2.0 ** -10 + dg * 2.0 ** -23) * aabs + dg * 2.0 ** -25 * float(v_sub.abs().max()) + 1e-30).all()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-3


def test_backward_product_reddit_like_f128_stated_tolerance(cuda_device, monkeypatch):
    """Round 6: the BACKWARD product dB = A^T dC at the headline size through the transposed handle that
    voltrix.autograd.SpMM builds (two-level side-car included), sampled rows of A^T (in-degree up to thousands) against
    torch.sparse.mm on the CPU with the window kernel's stated bound; and its step next to the forward's."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.autograd import SpMM

    indptr, indices = _graph("reddit_like")
    n, f = indptr.numel() - 1, 128
    op = SpMM(indptr, indices, n, hash_tag="stated_tolerance_backward_reddit")
    assert voltrix.two_level_of(op.handle_t[1]) is not None, "A^T of this graph takes the two-level format as well"
    grad_out = _random_fp16(n, f, 47)
    out = voltrix.spmm(*op.handle_t, num_nodes=n, num_edges=indices.numel(), feat=grad_out)
    from voltrix.autograd import csr_transpose_device

    t_indptr, t_indices = csr_transpose_device(indptr, indices, n, n)
    rows = _sample_rows(t_indptr, n, 47)
    _assert_stated_tolerance(out[rows], _sub_problem(t_indptr, t_indices, rows, grad_out), "reddit_like backward (A^T, two-level)")

    def step_ms(handle):
        """Median of five event-timed batches of five calls (a wall clock around ten calls read 7.8 ms once for a 1.27 ms step:
        one allocator stall inside the loop)."""
        call = lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=indices.numel(), feat=grad_out)  # noqa: E731
        for _ in range(3):
            call()
        times = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                call()
            e.record()
            e.synchronize()
            times.append(s.elapsed_time(e) / 5)
        return sorted(times)[2]

    fwd, bwd = step_ms(op.handle), step_ms(op.handle_t)
    print(f"reddit-like F=128 fp16: forward {fwd:.3f} ms, backward (A^T) {bwd:.3f} ms")
    assert bwd <= 1.3 * fwd + 0.05, (fwd, bwd)
