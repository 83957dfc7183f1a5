"""GPU: CSR -> (pointer1, hspa_packed, hind) is BIT-EXACT against the oracle for both preprocess routes
(fused GPU kernels through the C-ABI; the reference's three-stage pipeline through the JIT launch wrappers)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import synth_graphs
import voltrix
from oracle import oracle_c

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(11)
    out = {}
    out["single_row"] = (np.array([0, 1]), np.array([0]), 1)
    out["no_edges"] = (np.zeros(50, dtype=np.int64), np.array([], dtype=np.int64), 49)
    out["fifteen"] = (np.arange(16), rng.integers(0, 15, 15), 15)
    out["sixteen"] = (np.arange(17), rng.integers(0, 16, 16), 16)
    out["seventeen"] = (np.arange(18), rng.integers(0, 17, 17), 17)
    # duplicates + unsorted rows: a (row, col) pair counts once (bitmap), order inside a row is irrelevant
    out["dups_unsorted"] = (np.array([0, 4, 4, 9, 9] + [9] * 30), np.array([5, 1, 5, 1, 33, 2, 2, 0, 33]), 34)
    a = sp.random(700, 700, density=0.05, format="csr", random_state=7)
    out["sp700"] = (a.indptr, a.indices, 700)
    # one window with more than 8192 edges (LDS sort capacity) next to tiny ones -> global-memory sort path; its 9000
    # distinct columns = 1125 TC blocks also exceed the bitmap path's LDS stage (1024 blocks) -> global atomics there
    deg = np.zeros(96, dtype=np.int64)
    deg[16:32] = 700
    deg[40] = 3
    deg[95] = 9000
    n = 12000
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg if d > 0])
    out["huge_window"] = (indptr, indices, 96)
    return out


CASES = _cases()


def _check(handle, indptr, indices, n):
    p1, packed, hind = handle
    torch.cuda.synchronize()
    op1, opacked, ohind = oracle_c.csr_preprocess(np.asarray(indptr, np.int32), np.asarray(indices, np.int32), n)
    assert p1.dtype == torch.int32 and packed.dtype == torch.uint32 and hind.dtype == torch.int32
    assert p1.is_cuda and packed.is_cuda and hind.is_cuda
    assert np.array_equal(p1.cpu().numpy(), op1)
    assert np.array_equal(hind.cpu().numpy(), ohind)
    assert np.array_equal(packed.cpu().numpy(), opacked)


ROUTES = ["fused-sort", "fused-bitmap", "fused-mixed", "fused-auto", "reference"]


def _set_route(monkeypatch, route):
    """fused-sort / fused-bitmap / fused-mixed force one of the rank algorithms of the fused preprocess
    (VOLTRIX_PREPROCESS=fused:<path>, which the operator hands to the library as the ``path`` ARGUMENT of its entry points;
    mixed = windows up to 8192 edges sorted, bigger ones through the bitmap kernels)."""
    if route in ("fused-sort", "fused-bitmap", "fused-mixed"):
        monkeypatch.setenv("VOLTRIX_PREPROCESS", "fused:" + route.split("-")[1])
    else:
        monkeypatch.setenv("VOLTRIX_PREPROCESS", route.split("-")[0])


@pytest.mark.parametrize("route", ROUTES)
def test_fixtures_bit_exact(cuda_device, csr_fixture, route, monkeypatch):
    _set_route(monkeypatch, route)
    g = csr_fixture
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    _check(handle, g["indptr"], g["indices"], n)
    for got, want in zip(handle, (g["pointer1"], g["hspa_packed"], g["hind"])):
        assert np.array_equal(got.cpu().numpy(), want)  # the committed golden handle itself


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_cases_bit_exact(cuda_device, name, route, monkeypatch):
    _set_route(monkeypatch, route)
    indptr, indices, n = CASES[name]
    handle = voltrix.csr_preprocess(torch.as_tensor(np.asarray(indptr), dtype=torch.int32),
                                    torch.as_tensor(np.asarray(indices), dtype=torch.int32), n)
    _check(handle, indptr, indices, n)


@pytest.mark.parametrize("path", ["sort", "bitmap", "mixed"])
def test_non_square_universe_and_unstaged_window(cuda_device, path, monkeypatch):
    """96 rows x 12000 columns: with the universe declared, the bitmap path takes its global-atomics branch for the
    1125-block window; without it (default universe = num_nodes) out-of-universe ids are detected and the sort path
    redoes the work -- both give the oracle's bytes."""
    indptr, indices, n = CASES["huge_window"]
    ip = torch.as_tensor(np.asarray(indptr), dtype=torch.int32).cuda()
    ix = torch.as_tensor(np.asarray(indices), dtype=torch.int32).cuda()
    for num_cols in (12000, None, 0):
        handle = voltrix.csr_fused_preprocess_kernel(ip, ix, n, num_cols=num_cols, path=path)[:3]
        _check(handle, indptr, indices, n)


def test_bitmap_path_with_several_column_ranges(cuda_device, monkeypatch):
    """Universe of 1.3 M columns = 3 bitmap ranges of 2^19: columns clustered so that some ranges are empty for some
    windows, partial TC blocks straddle range boundaries, one window exceeds the LDS stage inside a single range and
    one window's columns all sit in the first range (pending partial block flushed at the end)."""
    rng = np.random.default_rng(5)
    ncols, nrows = 1_300_000, 80
    deg = rng.integers(0, 60, nrows)
    deg[0:16] = 700          # window 0: 11200 edges over all ranges (> 1024 blocks)
    deg[16:32] = 0           # window 1: empty
    rows = []
    for r, d in enumerate(deg):
        if 32 <= r < 48:     # window 2: first range only, odd number of distinct columns overall
            rows.append(np.sort(rng.choice(500_000, d, replace=False)))
        elif 48 <= r < 64:   # window 3: straddles the 2^19 and 2^20 boundaries tightly
            rows.append(np.sort(rng.choice(np.arange(524_288 - 40, 524_288 + 40), min(d, 60), replace=False)))
        else:
            rows.append(np.sort(rng.choice(ncols, d, replace=False)))
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in rows])])
    indices = np.concatenate(rows)
    ip = torch.as_tensor(indptr, dtype=torch.int32).cuda()
    ix = torch.as_tensor(indices, dtype=torch.int32).cuda()
    handle = voltrix.csr_fused_preprocess_kernel(ip, ix, nrows, num_cols=ncols, path="bitmap")[:3]
    _check(handle, indptr, indices, nrows)
    for path in ("sort", "mixed"):   # mixed: window 0 (11200 edges) through the bitmap kernels, the others sorted
        handle = voltrix.csr_fused_preprocess_kernel(ip, ix, nrows, num_cols=ncols, path=path)[:3]
        _check(handle, indptr, indices, nrows)
    # 3 ranges: the default choice is the mixed path
    handle = voltrix.csr_fused_preprocess_kernel(ip, ix, nrows, num_cols=ncols)[:3]
    _check(handle, indptr, indices, nrows)


def test_low_level_kernel_wrappers_like_reference_test(cuda_device):
    """tests/test_spmm_kernel.py:45-113 flow with the four public *_kernel wrappers and caller-allocated buffers
    (hspa / hind pre-filled with garbage: the kernels must write every element)."""
    np.random.seed(20)
    n = 2048
    a = sp.random(n, n, density=0.01, format="csr")
    indptr = torch.tensor(a.indptr, dtype=torch.int32)
    indices = torch.tensor(a.indices, dtype=torch.int32)
    e, w = indices.numel(), (n + 15) // 16
    e2c, e2r = torch.zeros(e, dtype=torch.int32), torch.zeros(e, dtype=torch.int32)
    bp, p1 = torch.zeros(w, dtype=torch.int32), torch.zeros(w + 1, dtype=torch.int32)
    voltrix.preprocess_kernel(edge_list=indices, node_pointer=indptr, block_partition=bp, edge_to_column=e2c,
                              edge_to_row=e2r, pointer1=p1)
    obp, oe2c, oe2r, op1 = oracle_c.preprocess(a.indptr, a.indices, n)
    assert np.array_equal(bp.numpy(), obp) and np.array_equal(p1.numpy(), op1)
    assert np.array_equal(e2c.numpy(), oe2c) and np.array_equal(e2r.numpy(), oe2r)
    t = int(p1[-1])
    hspa = torch.full((t * 128,), float("nan"), device="cuda")
    hind = torch.full((t * 8,), -7, dtype=torch.int32, device="cuda")
    packed = torch.full((t * 4,), 0x7FFFFFFF, dtype=torch.int32, device="cuda").view(torch.uint32)
    voltrix.hmat_gen_kernel(node_pointer=indptr.cuda(), edge_list=indices.cuda(), block_partition=bp.cuda(),
                            edge_to_column=e2c.cuda(), edge_to_row=e2r.cuda(), pointer1=p1.cuda(), hspa=hspa, hind=hind)
    voltrix.hmat_packed_swizzle_kernel(block_partition=bp.cuda(), pointer1=p1.cuda(), hspa=hspa, hspa_packed=packed)
    torch.cuda.synchronize()
    ohspa, ohind = oracle_c.hmat_gen(a.indptr, a.indices, obp, oe2c, oe2r, op1, n)
    assert np.array_equal(hspa.cpu().numpy(), ohspa) and np.array_equal(hind.cpu().numpy(), ohind)
    assert np.array_equal(packed.cpu().numpy(), oracle_c.hmat_packed_swizzle(op1, ohspa))


@pytest.mark.parametrize("path", ["sort", "bitmap", "mixed"])
def test_fused_preprocess_mid_size_with_large_windows(cuda_device, path, monkeypatch):
    indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda", scale=0.02)
    n = indptr.numel() - 1
    assert int((indptr[16::16] - indptr[:-16:16]).max()) > 8192  # exercises the global-memory sort
    p1, packed, hind, bp = voltrix.csr_fused_preprocess_kernel(indptr, indices, n, path=path)
    _check((p1, packed, hind), indptr.cpu().numpy(), indices.cpu().numpy(), n)
    assert np.array_equal(bp.cpu().numpy(), np.diff(p1.cpu().numpy()))


def test_mixed_path_power_law_windows(cuda_device, monkeypatch):
    """Power-law stand-in at 1/64 size with its full-size density (62.5 k rows, Zipf degrees: windows from a few hundred to
    tens of thousands of edges) over a 1.2 M-column universe (3 bitmap ranges): the default choice is the mixed path --
    wave sort, workgroup LDS sort and bitmap kernels all take windows -- and must give the oracle's bytes."""
    n, ncols = 62_500, 1_200_000
    g = torch.Generator(device="cuda").manual_seed(12)
    deg = synth_graphs.zipf_degrees(n, 400.0, 2.0, ncols // 4, g, torch.device("cuda"))
    total = int(deg.sum())
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    cols = torch.randint(0, ncols, (total,), device="cuda", generator=g)
    key = torch.unique(rows * ncols + cols)            # sorted, duplicate-free per row
    rows, cols = key // ncols, key % ncols
    indptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    indptr[1:] = torch.bincount(rows, minlength=n).cumsum(0)
    ip, ix = indptr.to(torch.int32), cols.to(torch.int32)
    per_window = ip[16::16] - ip[:-16:16]
    assert int(per_window.max()) > 8192 and int(per_window.min()) < 2048 and int(((per_window > 2048) & (per_window <= 8192)).sum()) > 0
    p1, packed, hind, _ = voltrix.csr_fused_preprocess_kernel(ip, ix, n, num_cols=ncols)
    _check((p1, packed, hind), ip.cpu().numpy(), ix.cpu().numpy(), n)


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("path", ["sort", "mixed"])
def test_bucket_ranking_mid_size_windows_and_clustered_columns(cuda_device, path, shuffle, monkeypatch):
    """Windows of 2049 .. 8192 edges take the bucket-ranking kernels (csr_bucket_count / _fill_kernel).  Seven windows over
    a 2 M-column universe: uniform columns; a tight band; 3000 edges on one column + singletons; exactly 8192 edges;
    exactly 2049 edges; a window whose 6000 edges sit on 900 neighbouring columns (the clustering test hands it on to
    the workgroup sort); and one of 11200 edges (global-memory sort, or the bitmap kernels on range-grouped keys).  Oracle bytes on every route."""
    rng = np.random.default_rng(77)
    ncols = 2_000_000

    def window(row_cols):
        return [np.sort(np.asarray(c, np.int64)) for c in row_cols]

    rows = []
    rows += window([rng.choice(ncols, 300, replace=False) for _ in range(16)])                       # 4800 uniform
    rows += window([700_000 + rng.choice(5000, 250, replace=False) for _ in range(16)])              # 4000 in a band
    rows += window([np.concatenate([[123_456], rng.choice(ncols, 187, replace=False)]) for _ in range(16)])  # one hot column
    rows += window([rng.choice(ncols, 512, replace=False) for _ in range(16)])                       # 8192 exactly
    rows += window([rng.choice(ncols, 129 if r == 0 else 128, replace=False) for r in range(16)])    # 2049 exactly
    rows += window([1_500_000 + rng.choice(900, 375, replace=False) for _ in range(16)])             # 6000 on 900 columns
    rows += window([rng.choice(ncols, 700, replace=False) for _ in range(16)])                       # 11200: above the LDS keys
    rows = [np.unique(r) for r in rows]
    if shuffle:   # rows need not be sorted (the reference condenses through a std::map)
        rows = [rng.permutation(r) for r in rows]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    indices = np.concatenate(rows).astype(np.int32)
    n = len(rows)
    per_window = np.diff(indptr[::16])
    assert per_window[3] == 8192 and per_window[4] == 2049 and per_window[6] > 8192 and (per_window > 2048).all()
    ip, ix = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    p1, packed, hind, _ = voltrix.csr_fused_preprocess_kernel(ip, ix, n, num_cols=ncols, path=path)
    _check((p1, packed, hind), indptr, indices, n)


FAMILY_SCALES = {"cora_like": 1.0, "reddit_like": 0.015, "products_like": 0.012, "protein_like": 0.05, "amazon0505_like": 0.3,
                 "amazon0601_like": 0.4, "com_amazon_like": 0.6, "dd_like": 0.8, "yeast_like": 0.4, "yeasth_like": 0.2,
                 "web_berkstan_like": 0.2, "ppi_like": 1.0, "ddi_like": 0.6, "fraud_yelp_rsr_like": 0.25}


@pytest.mark.parametrize("shuffled", [False, True])
@pytest.mark.parametrize("family", sorted(FAMILY_SCALES))
def test_every_graph_family_bit_exact(cuda_device, family, shuffled):
    """Round 6: one mid-size graph (0.01-2 M edges) of every stand-in family -- band, community, union of small graphs, Zipf rows with
    hub windows, dense blocks -- in its generating order and with shuffled labels (the windows of a shuffled graph share nothing:
    the other extreme of TC-block fill), through the operator's default device route: the oracle's bytes."""
    indptr, indices, _ = synth_graphs.generate(family, device="cuda", scale=FAMILY_SCALES[family])
    n = indptr.numel() - 1
    if shuffled:
        indptr, indices, _ = synth_graphs.shuffle_labels(indptr, indices, 31)
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    _check(handle, indptr.cpu().numpy(), indices.cpu().numpy(), n)
