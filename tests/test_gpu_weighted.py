"""GPU: weighted SpMM (value plane beside the bitmaps, WEIGHTED tiles of the window kernel) against the reference's
oracle expression with real values: ``torch.sparse_csr_tensor(indptr, indices, values) @ feat`` on the CPU.

Tolerance: A values and B are both rounded to the 16-bit operand type before the MFMA (fp32 accumulate), so against the
fp32 oracle the element-wise bound is (2^-10 + deg 2^-23) (|A| |B|) (two roundings of 2^-11 each, first order) plus the
accumulation term; against the oracle evaluated on the SAME rounded operands only deg 2^-23 (|A| |B|) remains."""
import numpy as np
import pytest
import torch

import voltrix
from conftest import load_csr_fixture
from test_hybrid_plan import _random_csr
from voltrix.weighted import value_plane

pytestmark = pytest.mark.gpu


def _oracle(indptr, indices, values, feat, n, ncols):
    a = torch.sparse_csr_tensor(torch.as_tensor(indptr), torch.as_tensor(indices), values.double(), size=(n, ncols))
    return (a @ feat.double())


def test_value_plane_follows_the_format_definition(cuda_device, csr_fixture):
    g = csr_fixture
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    values = torch.arange(1, len(indices) + 1, dtype=torch.float32)
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    plane = h.values32.cpu().numpy()
    hind = h.hind.cpu().numpy().reshape(-1, 8)
    p1 = h.blk_offsets.cpu().numpy()
    want = np.zeros_like(plane)
    for r in range(n):
        w = r // 16
        for e in range(g["indptr"][r], g["indptr"][r + 1]):
            c = g["indices"][e]
            blocks = hind[p1[w]:p1[w + 1]]
            cols = sorted(set(g["indices"][g["indptr"][16 * w]:g["indptr"][min(n, 16 * w + 16)]].tolist()))
            q = cols.index(c)
            assert blocks[q // 8, q % 8] == c
            want[p1[w] + q // 8, r % 16, q % 8] += values[e].item()      # duplicates add
    assert np.array_equal(plane, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("width", [32, 128, 200])
def test_weighted_spmm_matches_the_oracle(cuda_device, dtype, width, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    torch.manual_seed(3)
    values = torch.randn(len(indices))
    feat32 = torch.randn(n, width)
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    out = voltrix.spmm_weighted(h, feat32.to(dtype).cuda(), hash_tag=f"weighted_{width}")
    assert out.shape == (n, width) and out.dtype == torch.float32 and not torch.isnan(out).any()
    ref = _oracle(g["indptr"], g["indices"], values, feat32.to(dtype) if dtype != torch.float32 else feat32, n, n)
    absref = _oracle(g["indptr"], g["indices"], values.abs(), feat32.abs(), n, n)
    deg = torch.from_numpy(np.diff(g["indptr"]).astype(np.float64))
    u = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    bound = (u + (deg[:, None] + 1) * 2.0 ** -23) * absref + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    assert float((out.cpu().double() - ref).norm() / ref.norm()) < (2e-2 if dtype == torch.bfloat16 else 2e-3)
    # same rounded operands on both sides: accumulation order only
    vdt = torch.float16 if dtype == torch.float32 else dtype
    if dtype != torch.float32:
        ref_same = _oracle(g["indptr"], g["indices"], values.to(vdt).float(), feat32.to(dtype).float(), n, n)
        assert ((out.cpu().double() - ref_same).abs() <= (deg[:, None] + 1) * 2.0 ** -23 * absref + 1e-6).all()
    assert torch.equal(out, voltrix.spmm_weighted(h, feat32.to(dtype).cuda()))


def test_weighted_with_unit_values_is_the_binary_product_and_duplicates_add(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    indptr, indices = _random_csr(900, 60, seed=4)
    ip, ix = torch.from_numpy(indptr), torch.from_numpy(indices)
    feat = torch.randint(-3, 4, (900, 64)).half().cuda()
    h = voltrix.csr_preprocess_weighted(ip, ix, torch.ones(len(indices)), 900)
    plain = voltrix.csr_preprocess(ip, ix, 900)
    plain[1].hash_tag = "weighted_vs_binary"
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((h.blk_offsets, h.hspa_packed, h.hind), plain))
    assert torch.equal(voltrix.spmm_weighted(h, feat, hash_tag="weighted_unit"),
                       voltrix.spmm(*plain, num_nodes=900, num_edges=len(indices), feat=feat))
    # duplicate entries: values add (torch.sparse.mm semantics), the pattern counts them once
    dup_indices = torch.cat([ix[:3], ix])          # row 0 gets its first three columns twice (if it has three)
    if indptr[1] >= 3:
        dup_indptr = ip.clone()
        dup_indptr[1:] += 3
        vals = torch.ones(len(dup_indices))
        hd = voltrix.csr_preprocess_weighted(dup_indptr, dup_indices, vals, 900)
        out = voltrix.spmm_weighted(hd, torch.ones(900, 8).half().cuda(), hash_tag="weighted_dup")
        assert float(out[0, 0]) == float(indptr[1]) + 3.0


def test_value_plane_in_chunks_and_without_an_fp32_master(cuda_device, monkeypatch):
    """Round 5: the plane is built window chunk by window chunk (bounded temporaries) -- the same bits as in one piece -- and a
    handle too large for an fp32 master keeps only the 16-bit plane it was built with."""
    from voltrix import weighted

    indptr_np, indices_np = _random_csr(2500, 40, seed=11)
    indptr, indices = torch.from_numpy(indptr_np), torch.from_numpy(indices_np)
    n = 2500
    torch.manual_seed(5)
    values = torch.randn(len(indices_np))
    whole = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    for chunk in (1, 1000, 7777):
        monkeypatch.setattr(weighted, "CHUNK_EDGES", chunk)
        again = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
        assert torch.equal(again.values32, whole.values32), chunk
    monkeypatch.setattr(weighted, "MASTER_PLANE_MAX_BYTES", 0)
    lean = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    assert lean.values32 is None and list(lean.planes) == [torch.float16]
    assert torch.equal(lean.planes[torch.float16], whole.values32.half())
    feat = torch.randn(n, 64, device=cuda_device).half()
    assert torch.equal(voltrix.spmm_weighted(lean, feat, hash_tag="lean"), voltrix.spmm_weighted(whole, feat, hash_tag="lean"))
    with pytest.raises(AssertionError, match="without an fp32 master"):
        voltrix.spmm_weighted(lean, feat.bfloat16())
    lean_bf = voltrix.csr_preprocess_weighted(indptr, indices, values, n, plane_dtype=torch.bfloat16)
    assert list(lean_bf.planes) == [torch.bfloat16] and torch.equal(lean_bf.planes[torch.bfloat16], whole.values32.bfloat16())


# ---- round 6: values that factor as r_i c_j run on the BINARY operator between two row scalings (no value plane) ----------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_scale_rows_kernel_matches_torch(cuda_device, dtype):
    from voltrix import capi

    torch.manual_seed(0)
    x = torch.randn(1003, 136, device=cuda_device).to(dtype)
    f = (torch.rand(1003, device=cuda_device) * 4 - 2).float()
    out = torch.empty_like(x)
    capi.launch_scale_rows(x, f, out, torch.cuda.current_stream().cuda_stream)
    want = (x.float() * f[:, None]).to(dtype)
    assert torch.equal(out.view(torch.int16 if dtype != torch.float32 else torch.int32),
                       want.view(torch.int16 if dtype != torch.float32 else torch.int32))
    capi.launch_scale_rows(x, f, x, torch.cuda.current_stream().cuda_stream)          # in place
    assert torch.equal(x.view(torch.int16 if dtype != torch.float32 else torch.int32),
                       want.view(torch.int16 if dtype != torch.float32 else torch.int32))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("kind", ["symmetric", "row_mean", "random_factors", "stated"])
def test_separable_values_run_on_the_binary_operator_and_match_the_oracle(cuda_device, dtype, kind, monkeypatch):
    """v_ij = r_i c_j (GCN's D^-1/2 A D^-1/2, the mean aggregator's D^-1 A, arbitrary positive factors): detected exactly, no value
    plane is built, and the product obeys a TIGHTER bound than the value-plane path (one 16-bit rounding, of c_j b_jk, instead of
    two): |out - ref| <= (u + (deg + 2) 2^-23) (|A| |B|)."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix import weighted

    monkeypatch.setattr(weighted, "separable_pays", lambda *a: True)   # the scalings whatever the sizes say (this fixture is tiny)
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    deg = torch.from_numpy(np.diff(g["indptr"]).astype(np.int64))
    rows = torch.repeat_interleave(torch.arange(n), deg)
    indeg = torch.bincount(indices.long(), minlength=n).double().clamp(min=1)
    torch.manual_seed(8)
    if kind == "symmetric":
        r, c = deg.double().clamp(min=1).rsqrt(), indeg.rsqrt()
    elif kind == "row_mean":
        r, c = 1.0 / deg.double().clamp(min=1), torch.ones(n, dtype=torch.float64)
    else:
        r, c = torch.rand(n, dtype=torch.float64) * 3 + 0.05, torch.rand(n, dtype=torch.float64) * 2 + 0.1
    values = (r[rows] * c[indices.long()]).float()
    if kind == "stated":
        h = voltrix.csr_preprocess_weighted(indptr, indices, None, n, row_scale=r.float(), col_scale=c.float())
    else:
        h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    assert h.separable and h.values32 is None and not h.planes
    feat32 = torch.randn(n, 72)
    feat = feat32.to(dtype)
    out = voltrix.spmm_weighted(h, feat.cuda(), hash_tag=f"separable_{kind}")
    assert out.shape == (n, 72) and out.dtype == torch.float32
    ref = _oracle(g["indptr"], g["indices"], values, feat if dtype != torch.float32 else feat32, n, n)
    absref = _oracle(g["indptr"], g["indices"], values.abs(), feat32.abs(), n, n)
    u = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11        # ONE rounding of the scaled operand (fp32 features: to fp16)
    if kind != "stated":
        u += 2.0 ** -13                                             # the detection's tolerance on r_i c_j itself
    bound = (u + (deg.double()[:, None] + 2) * 2.0 ** -23) * absref + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    # the general value plane computes the same product (two roundings): the two paths agree within the sum of their bounds
    plane = voltrix.csr_preprocess_weighted(indptr, indices, values, n, separable=False)
    assert not plane.separable and plane.values32 is not None
    out_plane = voltrix.spmm_weighted(plane, feat.cuda(), hash_tag=f"separable_{kind}_plane")
    u2 = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((out.cpu().double() - out_plane.cpu().double()).abs() <= (u + u2 + (2 * deg.double()[:, None] + 4) * 2.0 ** -23) * absref + 2e-6).all()


def test_non_separable_and_duplicate_values_keep_the_value_plane(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    torch.manual_seed(1)
    assert not voltrix.csr_preprocess_weighted(indptr, indices, torch.rand(len(indices)) + 0.5, n).separable     # random positive
    assert not voltrix.csr_preprocess_weighted(indptr, indices, torch.randn(len(indices)), n).separable         # signs
    with pytest.raises(AssertionError, match="do not factor"):
        voltrix.csr_preprocess_weighted(indptr, indices, torch.rand(len(indices)) + 0.5, n, separable=True)
    # duplicate (row, col) entries ADD in the weighted product and count once in the binary format: never separable
    ip, ix = _random_csr(300, 20, seed=2)
    ip, ix = torch.from_numpy(ip), torch.from_numpy(ix)
    if int(ip[1]) >= 2:
        dup_ix = torch.cat([ix[:2], ix])
        dup_ip = ip.clone()
        dup_ip[1:] += 2
        assert not voltrix.csr_preprocess_weighted(dup_ip, dup_ix, torch.ones(len(dup_ix)), 300).separable


@pytest.mark.parametrize("kind", ["symmetric", "general"])
def test_weighted_autograd_gradient_is_the_transposed_weighted_product(cuda_device, kind, monkeypatch):
    """d/dB of sum(w * (A_values B)) = A_values^T w, separable values (both directions on the binary operator, factors swapped) and
    general ones (value planes of A and A^T) against torch.sparse on the CPU."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.autograd import SpMM

    ip_np, ix_np = _random_csr(700, 30, seed=6)
    ip, ix = torch.from_numpy(ip_np), torch.from_numpy(ix_np)
    n = 700
    deg = torch.from_numpy(np.diff(ip_np).astype(np.int64))
    rows = torch.repeat_interleave(torch.arange(n), deg)
    torch.manual_seed(4)
    if kind == "symmetric":
        indeg = torch.bincount(ix.long(), minlength=n).double().clamp(min=1)
        values = (deg.double().clamp(min=1).rsqrt()[rows] * indeg.rsqrt()[ix.long()]).float()
    else:
        values = torch.randn(len(ix_np))
    op = SpMM(ip, ix, n, values=values, hash_tag=f"autograd_weighted_{kind}")
    assert op.weighted.separable == (kind == "symmetric") == op.weighted_t.separable
    feat = torch.randn(n, 40, device=cuda_device).half().requires_grad_(True)
    w = torch.randn(n, 40, device=cuda_device)
    out = op(feat)
    (out * w).sum().backward()
    a = torch.sparse_csr_tensor(ip, ix, values.double(), size=(n, n))
    ref_out = a @ feat.detach().cpu().double()
    ref_grad = a.to_dense().T @ w.cpu().double()
    scale_o = (torch.sparse_csr_tensor(ip, ix, values.abs().double(), size=(n, n)) @ feat.detach().cpu().abs().double())
    scale_g = (torch.sparse_csr_tensor(ip, ix, values.abs().double(), size=(n, n)).to_dense().T @ w.cpu().abs().double())
    assert ((out.detach().cpu().double() - ref_out).abs() <= 2.0 ** -9 * scale_o + 1e-5).all()
    # the gradient passes through fp16 twice more: the incoming gradient is rounded like any operand, the result is cast to feat's dtype
    assert ((feat.grad.cpu().double() - ref_grad).abs() <= 2.0 ** -8 * scale_g + 2.0 ** -10 * ref_grad.abs() + 1e-4).all()


def test_separable_handle_takes_the_value_plane_where_the_row_scalings_would_cost_more(cuda_device, monkeypatch):
    """2 N F (s + 4) bytes of row scalings against 256 bytes per TC block: a handle of short windows multiplied by a WIDE operand is
    better off with the plane (papers-like x 128: 94 ms separable, 80.5 with the plane), built lazily from the kept CSR -- same product."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix import weighted

    g = load_csr_fixture("cora_like")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    deg = torch.from_numpy(np.diff(g["indptr"]).astype(np.int64))
    rows = torch.repeat_interleave(torch.arange(n), deg)
    values = (1.0 / deg.double().clamp(min=1))[rows].float()
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n)
    assert h.separable and not h.planes
    assert weighted.separable_pays(h, 8, 2) and not weighted.separable_pays(h, 512, 2)      # narrow: scalings; wide: the plane
    feat = torch.randn(n, 512).half()
    out = voltrix.spmm_weighted(h, feat.cuda(), hash_tag="separable_lazy_plane")
    assert torch.float16 in h.planes                                                          # built on demand, once
    ref = _oracle(g["indptr"], g["indices"], values, feat, n, n)
    absref = _oracle(g["indptr"], g["indices"], values.abs(), feat.float().abs(), n, n)
    assert ((out.cpu().double() - ref).abs() <= (2.0 ** -10 + (deg.double()[:, None] + 1) * 2.0 ** -23) * absref + 1e-6).all()
    narrow = voltrix.spmm_weighted(h, feat[:, :8].contiguous().cuda())                       # the separable path on the same handle
    assert ((narrow.cpu().double() - ref[:, :8]).abs() <= (2.0 ** -10 + (deg.double()[:, None] + 2) * 2.0 ** -23) * absref[:, :8] + 1e-6).all()


def test_prescaled_and_unscaled_forms_of_the_separable_product(cuda_device, monkeypatch):
    """Callers that fold the factors into their own kernels: prescaled=True skips the pass over B, postscale=False the pass over C;
    composing them by hand gives the same bits as the full call."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix import weighted

    monkeypatch.setattr(weighted, "separable_pays", lambda *a: True)
    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    indptr, indices = torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"])
    torch.manual_seed(3)
    r, c = torch.rand(n) + 0.5, torch.rand(n) + 0.5
    h = voltrix.csr_preprocess_weighted(indptr, indices, None, n, row_scale=r, col_scale=c)
    feat = torch.randn(n, 64, device=cuda_device).half()
    full = voltrix.spmm_weighted(h, feat, hash_tag="separable_forms")
    pre = (feat.float() * h.col_scale[:, None]).half()
    by_hand = voltrix.spmm_weighted(h, pre, prescaled=True, postscale=False) * h.row_scale[:, None]
    assert torch.equal(full, by_hand)
    plane = voltrix.csr_preprocess_weighted(indptr, indices, torch.ones(len(indices)), n, separable=False)
    with pytest.raises(AssertionError, match="separable values only"):
        voltrix.spmm_weighted(plane, feat, prescaled=True)


# ---- round 6: new values on the same pattern (attention coefficients, trained edge weights): one scatter, no rebuild ------------------
@pytest.mark.parametrize("duplicates", [False, True])
def test_update_values_equals_a_rebuilt_handle_bit_for_bit(cuda_device, duplicates, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix import weighted

    ip_np, ix_np = _random_csr(900, 35, seed=21)
    if duplicates:      # the first entry of every non-empty row once more: the two values ADD in the weighted product
        rows = [ix_np[ip_np[r]:ip_np[r + 1]] for r in range(900)]
        rows = [np.concatenate([r[:1], r]) for r in rows]
        ip_np = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        ix_np = np.concatenate(rows).astype(np.int32)
    ip, ix = torch.from_numpy(ip_np), torch.from_numpy(ix_np)
    n = 900
    torch.manual_seed(8)
    v1, v2, v3 = torch.randn(len(ix_np)), torch.randn(len(ix_np)), torch.rand(len(ix_np)) + 0.1
    h = voltrix.csr_preprocess_weighted(ip, ix, v1, n, plane_dtype=torch.float16, separable=False)
    only_master = voltrix.csr_preprocess_weighted(ip, ix, v1, n, separable=False)
    feat = torch.randn(n, 72, device=cuda_device).half()
    for v in (v2, v3):
        assert voltrix.update_edge_values(h, v) is h
        fresh = voltrix.csr_preprocess_weighted(ip, ix, v, n, separable=False)
        assert h.slot_duplicates == duplicates
        # duplicates add in the fp32 master (kept); without them the 16-bit planes are written directly and the master is dropped:
        # a plane of another 16-bit type is then made from the latest values through the same edge -> plane map
        assert (h.values32 is not None) == duplicates
        assert torch.equal(h.planes[torch.float16], fresh.values32.half())
        voltrix.spmm_weighted(h, feat.bfloat16(), hash_tag=f"upd{duplicates}")
        assert torch.equal(h.planes[torch.bfloat16], fresh.values32.bfloat16())
        voltrix.update_edge_values(only_master, v)          # no 16-bit plane yet: the master itself is rewritten
        assert not only_master.planes and torch.equal(only_master.values32, fresh.values32)
        out = voltrix.spmm_weighted(h, feat)
        assert torch.equal(out, voltrix.spmm_weighted(fresh, feat, hash_tag=f"upd{duplicates}"))
        ref = _oracle(ip_np, ix_np, v, feat.cpu(), n, n)
        scale = _oracle(ip_np, ix_np, v.abs(), feat.cpu().abs(), n, n)
        assert ((out.cpu().double() - ref).abs() <= 2.0 ** -9 * scale + 1e-5).all()
    # a handle without an fp32 master: the 16-bit plane is written directly (duplicate-free), refused with duplicates
    monkeypatch.setattr(weighted, "MASTER_PLANE_MAX_BYTES", 0)
    lean = voltrix.csr_preprocess_weighted(ip, ix, v1, n, separable=False)
    if duplicates:
        with pytest.raises(AssertionError, match="fp32 master"):
            voltrix.update_edge_values(lean, v2)
    else:
        voltrix.update_edge_values(lean, v2)
        assert torch.equal(lean.planes[torch.float16], voltrix.csr_preprocess_weighted(ip, ix, v2, n, separable=False).planes[torch.float16])


def test_update_values_on_separable_handles_and_through_autograd(cuda_device, monkeypatch):
    """Separable -> separable replaces the factors; separable -> general turns the handle into a value-plane one; autograd.SpMM updates
    both directions (the transposed plane through the transposed edge order)."""
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.autograd import SpMM

    ip_np, ix_np = _random_csr(600, 25, seed=23)
    ip, ix = torch.from_numpy(ip_np), torch.from_numpy(ix_np)
    n = 600
    deg = torch.from_numpy(np.diff(ip_np).astype(np.int64))
    rows = torch.repeat_interleave(torch.arange(n), deg)
    indeg = torch.bincount(ix.long(), minlength=n).double().clamp(min=1)
    sym = (deg.double().clamp(min=1).rsqrt()[rows] * indeg.rsqrt()[ix.long()]).float()
    mean = (1.0 / deg.double().clamp(min=1))[rows].float()
    torch.manual_seed(9)
    general = torch.rand(len(ix_np)) + 0.2
    feat = torch.randn(n, 48, device=cuda_device).half()

    def close(out, values):
        ref = _oracle(ip_np, ix_np, values, feat.cpu(), n, n)
        scale = _oracle(ip_np, ix_np, values.abs(), feat.cpu().abs(), n, n)
        return bool(((out.cpu().double() - ref).abs() <= 2.0 ** -9 * scale + 1e-5).all())

    h = voltrix.csr_preprocess_weighted(ip, ix, sym, n)
    assert h.separable and close(voltrix.spmm_weighted(h, feat, hash_tag="upd_sep"), sym)
    voltrix.update_edge_values(h, mean)
    assert h.separable and close(voltrix.spmm_weighted(h, feat), mean)
    voltrix.update_edge_values(h, general)
    assert not h.separable and h.values32 is not None and close(voltrix.spmm_weighted(h, feat), general)
    voltrix.update_edge_values(h, sym)          # a general handle stays general: the values go through the plane
    assert not h.separable and close(voltrix.spmm_weighted(h, feat), sym)

    for first, second in ((general, general * 0.5 + 0.1), (sym, mean), (sym, general)):
        op = SpMM(ip, ix, n, values=first, hash_tag="upd_autograd")
        op.update_values(second)
        b = feat.clone().requires_grad_(True)
        w = torch.randn(n, 48, device=cuda_device)
        out = op(b)
        (out * w).sum().backward()
        assert close(out.detach(), second)
        a = torch.sparse_csr_tensor(ip, ix, second.double(), size=(n, n)).to_dense()
        ref_grad = a.T @ w.cpu().double()
        scale_g = a.abs().T @ w.cpu().abs().double()
        assert ((b.grad.cpu().double() - ref_grad).abs() <= 2.0 ** -8 * scale_g + 2.0 ** -10 * ref_grad.abs() + 1e-4).all()


@pytest.mark.parametrize("relabel,method", [(True, "clusters"), (False, "bfs"), (True, "identity")])
def test_normalised_adjacency_on_a_reordered_handle(cuda_device, relabel, method, monkeypatch):
    """Round 6: reorder + separable edge values in one handle -- csr_preprocess_reordered(row_scale=, col_scale=) permutes the factors
    with the rows, spmm_reordered runs diag(r) A (diag(c) B) on the binary operator; against torch.sparse with the values, in the new
    order, un-permuted, and for a row-only handle (B and C in the caller's order)."""
    import synth_graphs

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    indptr, indices, _ = synth_graphs.generate("com_amazon_like", scale=0.05)
    n = indptr.numel() - 1
    indptr, indices, _ = synth_graphs.shuffle_labels(indptr, indices, 3)
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n), deg)
    indeg = torch.bincount(indices.long(), minlength=n)
    r = deg.double().clamp(min=1).rsqrt().float()
    c = indeg.double().clamp(min=1).rsqrt().float()
    values = r[rows] * c[indices.long()]
    h = voltrix.csr_preprocess_reordered(indptr, indices, n, method=method, relabel=relabel, row_scale=r, col_scale=c)
    assert h.relabelled == (relabel and method != "identity")
    for width in (64, 100):
        feat = torch.randn(n, width, device=cuda_device).half()
        ref = _oracle(indptr.numpy(), indices.numpy(), values, feat.cpu(), n, n)
        scale = _oracle(indptr.numpy(), indices.numpy(), values.abs(), feat.cpu().abs(), n, n)
        fin = voltrix.permute_features(h, feat)
        out_new = voltrix.spmm_reordered(h, fin, hash_tag=f"reordered_weighted_{relabel}_{method}")
        out = voltrix.unpermute_output(h, out_new)
        assert out.shape == (n, width) and ((out.cpu().double() - ref).abs() <= 2.0 ** -10 * scale + 1e-6).all()
        again = voltrix.spmm_reordered(h, fin, unpermute=True)
        assert torch.equal(again, out)
        assert torch.equal(fin, voltrix.permute_features(h, feat))        # the operand was not scaled in place


# ---- round 6: the CSR row-gather kernel with edge values (no plane) and the value-plane scatter kernel -----------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("width", [8, 72, 256, 520])
def test_csr_kernel_with_values_matches_the_oracle(cuda_device, dtype, width, monkeypatch):
    """voltrix_launch_spmm_csr_rows_weighted through spmm_weighted (VOLTRIX_CSR_PATH=1): fp32 values x rows as they are, one fused
    multiply-add per element -- only the operand's own rounding and the fp32 sum remain: (deg + 1) 2^-23 (|A| |B|) against the oracle on
    the same operand; duplicate entries add."""
    from voltrix import capi

    monkeypatch.setenv("VOLTRIX_CSR_PATH", "1")
    ip_np, ix_np = _random_csr(1100, 60, seed=31)
    rows = [ix_np[ip_np[r]:ip_np[r + 1]] for r in range(1100)]
    rows = [np.concatenate([r[:1], r]) if i % 7 == 0 else r for i, r in enumerate(rows)]          # some duplicates
    ip_np = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ix_np = np.concatenate(rows).astype(np.int32)
    ip, ix = torch.from_numpy(ip_np), torch.from_numpy(ix_np)
    n = 1100
    torch.manual_seed(12)
    values = torch.randn(len(ix_np))
    feat = torch.randn(n, width, device=cuda_device).to(dtype)
    h = voltrix.csr_preprocess_weighted(ip, ix, values, n, separable=False)
    out = voltrix.spmm_weighted(h, feat, hash_tag="weighted_csr")
    assert out.shape == (n, width) and out.dtype == torch.float32
    ref = _oracle(ip_np, ix_np, values, feat.cpu().float(), n, n)
    scale = _oracle(ip_np, ix_np, values.abs(), feat.cpu().float().abs(), n, n)
    deg = torch.from_numpy(np.diff(ip_np).astype(np.float64))[:, None]
    assert ((out.cpu().double() - ref).abs() <= (deg + 1) * 2.0 ** -23 * scale + 1e-30).all()
    # the entry point itself, without the XCD ranges
    raw = torch.empty(n, width, dtype=torch.float32, device=cuda_device)
    capi.launch_spmm_csr_rows(h.csr[0], h.csr[1], n, feat, raw, torch.cuda.current_stream().cuda_stream, 0, values=h.csr[2])
    assert ((raw.cpu().double() - ref).abs() <= (deg + 1) * 2.0 ** -23 * scale + 1e-30).all()


def test_exact_fp32_weighted_product_and_the_measured_path(cuda_device, monkeypatch):
    """VOLTRIX_FP32_MODE=exact on a weighted handle used to be refused (16-bit planes); it now runs the CSR kernel with values.  Without
    the flag a handle of short windows times both paths once per (width, dtype) and remembers the choice; either is within the plane
    path's bound."""
    import synth_graphs

    indptr, indices, _ = synth_graphs.generate("com_amazon_like", scale=0.2)
    n = indptr.numel() - 1
    torch.manual_seed(13)
    values = torch.rand(indices.numel()) + 0.1
    h = voltrix.csr_preprocess_weighted(indptr, indices, values, n, separable=False)
    feat = torch.randn(n, 64, device=cuda_device)
    ref = _oracle(indptr.numpy(), indices.numpy(), values, feat.cpu(), n, n)
    scale = _oracle(indptr.numpy(), indices.numpy(), values.abs(), feat.cpu().abs(), n, n)
    deg = (indptr[1:] - indptr[:-1]).double()[:, None]
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact")
    exact = voltrix.spmm_weighted(h, feat, hash_tag="weighted_exact")
    assert ((exact.cpu().double() - ref).abs() <= (deg + 1) * 2.0 ** -23 * scale + 1e-30).all()
    monkeypatch.delenv("VOLTRIX_FP32_MODE")
    for operand in (feat, feat.half()):
        out = voltrix.spmm_weighted(h, operand)
        key = (64, str(operand.dtype))
        assert h.path_choice[key] in ("csr", "plane")
        assert ((out.cpu().double() - ref).abs() <= (2.0 ** -9 + deg * 2.0 ** -23) * scale + 1e-6).all()
        assert torch.equal(out, voltrix.spmm_weighted(h, operand))                      # the remembered choice: the same kernel again
    monkeypatch.setenv("VOLTRIX_CSR_PATH", "0")
    plane = voltrix.spmm_weighted(h, feat.half())
    assert ((plane.cpu().double() - ref).abs() <= (2.0 ** -9 + deg * 2.0 ** -23) * scale + 1e-6).all()
    # new values: nothing to rebuild for the CSR path, one scatter for the plane -- both see them
    new = torch.rand(indices.numel()) + 0.5
    voltrix.update_edge_values(h, new)
    ref2 = _oracle(indptr.numpy(), indices.numpy(), new, feat.cpu(), n, n)
    scale2 = _oracle(indptr.numpy(), indices.numpy(), new.abs(), feat.cpu().abs(), n, n)
    assert ((voltrix.spmm_weighted(h, feat.half()).cpu().double() - ref2).abs() <= (2.0 ** -9 + deg * 2.0 ** -23) * scale2 + 1e-6).all()
    monkeypatch.setenv("VOLTRIX_CSR_PATH", "1")
    assert ((voltrix.spmm_weighted(h, feat.half()).cpu().double() - ref2).abs() <= (2.0 ** -10 + deg * 2.0 ** -23) * scale2 + 1e-6).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_scatter_values_kernel_matches_torch(cuda_device, dtype):
    from voltrix import capi

    torch.manual_seed(14)
    total, count = 1 << 20, 300_001
    slots = torch.randperm(total, device=cuda_device)[:count].contiguous()
    values = torch.randn(count, device=cuda_device) * 1e3
    values[:4] = torch.tensor([0.0, float("inf"), 3.4e38, 1e-40], device=cuda_device)      # zero, inf, near-overflow, subnormal
    plane = torch.full((total,), 7.0, dtype=dtype, device=cuda_device)
    want = plane.clone()
    want[slots] = values.to(dtype)
    capi.launch_scatter_values(values, slots, plane, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(plane, want)
    capi.launch_scatter_values(values[:0], slots[:0], plane, torch.cuda.current_stream().cuda_stream)           # empty: a no-op
    assert torch.equal(plane, want)


def test_the_handle_owns_its_edge_values(cuda_device, monkeypatch):
    """The CSR kernel with values reads the handle's values at every call: a caller that writes into its own tensor afterwards must not
    change the product (both paths keep computing the values of the last csr_preprocess_weighted / update_edge_values)."""
    ip_np, ix_np = _random_csr(500, 12, seed=41)
    ip, ix = torch.from_numpy(ip_np).cuda(), torch.from_numpy(ix_np).cuda()
    torch.manual_seed(15)
    mine = (torch.rand(len(ix_np), device=cuda_device) + 0.5).float()
    h = voltrix.csr_preprocess_weighted(ip, ix, mine, 500, separable=False)
    feat = torch.randn(500, 32, device=cuda_device).half()
    outs = {}
    for path in ("0", "1"):
        monkeypatch.setenv("VOLTRIX_CSR_PATH", path)
        outs[path] = voltrix.spmm_weighted(h, feat, hash_tag="owned_values")
    mine.mul_(3.0)
    for path in ("0", "1"):
        monkeypatch.setenv("VOLTRIX_CSR_PATH", path)
        assert torch.equal(voltrix.spmm_weighted(h, feat), outs[path])
    voltrix.update_edge_values(h, mine)
    tripled = mine.cpu().clone()
    mine.zero_()
    ref = _oracle(ip_np, ix_np, tripled, feat.cpu(), 500, 500)
    scale = _oracle(ip_np, ix_np, tripled.abs(), feat.cpu().abs(), 500, 500)
    for path in ("0", "1"):
        monkeypatch.setenv("VOLTRIX_CSR_PATH", path)
        assert ((voltrix.spmm_weighted(h, feat).cpu().double() - ref).abs() <= 2.0 ** -9 * scale + 1e-6).all()
