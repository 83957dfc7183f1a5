"""CPU: the oracle (numpy + C restatement) against every pin it has -- the reference's known answers
(SURVEY.md section 8c), the committed fixtures, an independent unpack round trip, and the reference's own test
criterion (calc_diff vs torch.sparse.mm on tests/test_spmm*.py inputs)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, load_csr_fixture
from oracle import oracle_c, oracle_np, torch_ref


def _load_sprandom_fixture(c):
    """Committed adjacency (row bitmaps) of one of the reference's own seeded test inputs -> CSR with sorted rows, exactly
    what ``np.random.seed(s); sp.random(N, N, density=d, format="csr")`` returns with the numpy / scipy of SURVEY.md 8c."""
    g = np.load(os.path.join(GOLDEN, f"sprandom_N{c['N']}_d{c['density']}_seed{c['seed']}.npz"))
    dense = np.unpackbits(g["bits"], axis=1)[:, : c["N"]].astype(bool)
    rows, cols = np.nonzero(dense)          # row-major: rows ascending, columns ascending inside a row
    indptr = np.zeros(c["N"] + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(np.bincount(rows, minlength=c["N"]))
    return indptr, cols.astype(np.int32)


def test_known_answers_from_reference_preprocess():
    """The reference's own ``voltrix::preprocess`` answers recorded in SURVEY.md section 8c (nnz / W / T / TC blocks per
    window) on the committed CSR arrays of its seeded test inputs -- no dependence on the installed sp.random stream."""
    ka = json.load(open(os.path.join(GOLDEN, "ref_known_answers.json")))
    for c in ka["cases"]:
        indptr, indices = _load_sprandom_fixture(c)
        assert len(indices) == c["nnz"] and int(indices.astype(np.int64).sum()) == c["indices_sum"]
        assert int(indptr.astype(np.int64).sum()) == c["indptr_sum"]
        bp, e2c, e2r, p1 = oracle_c.preprocess(indptr, indices, c["N"])
        assert len(bp) == c["W"]
        assert int(p1[-1]) == c["T"]
        assert int(bp.min()) == c["min"] and int(bp.max()) == c["max"]
        assert (e2r == np.repeat(np.arange(c["N"]), np.diff(indptr))).all()


def test_committed_fixtures_are_the_seeded_inputs_when_the_stream_matches():
    """Where the installed numpy / scipy still produce the recorded stream (they do in this image), the committed arrays
    ARE ``sp.random``'s output; elsewhere the committed arrays stand on their own (checksums above)."""
    ka = json.load(open(os.path.join(GOLDEN, "ref_known_answers.json")))
    c = ka["cases"][0]
    np.random.seed(c["seed"])
    a = sp.random(c["N"], c["N"], density=c["density"], format="csr")
    if a.nnz == c["nnz"] and int(a.indices.astype(np.int64).sum()) == c["indices_sum"]:
        indptr, indices = _load_sprandom_fixture(c)
        assert np.array_equal(indptr, a.indptr) and np.array_equal(indices, a.indices)


def test_oracle_np_equals_oracle_c_and_fixture(csr_fixture):
    g = csr_fixture
    n = int(g["num_nodes"])
    for impl in (oracle_np, oracle_c):
        bp, e2c, e2r, p1 = impl.preprocess(g["indptr"], g["indices"], n)
        assert (bp == g["block_partition"]).all() and (p1 == g["pointer1"]).all()
        assert (e2c == g["edge_to_column"]).all() and (e2r == g["edge_to_row"]).all()
        hspa, hind = impl.hmat_gen(g["indptr"], g["indices"], bp, e2c, e2r, p1, n)
        assert (hind == g["hind"]).all()
        assert (impl.hmat_packed_swizzle(p1, hspa) == g["hspa_packed"]).all()


def test_toy_empty_window_quirk():
    g = load_csr_fixture("toy40")
    # rows 16..31 have no edge: the reference still gives that window one (all-zero) TC block
    assert g["block_partition"].tolist() == [2, 1, 1]
    assert g["pointer1"].tolist() == [0, 2, 3, 4]
    # SURVEY.md section 8c, literally: what the reference's compiled preprocess printed for the toy case's 16 edges of
    # window 0 (rows 0, 1, 2 and the unsorted row 15) -- computed here by both restatements, not read from the fixture
    survey_edge_to_column = [3, 7, 11, 3, 0, 1, 2, 4, 5, 6, 8, 9, 10, 2, 1, 0]
    for impl in (oracle_np, oracle_c):
        bp, e2c, e2r, p1 = impl.preprocess(g["indptr"], g["indices"], int(g["num_nodes"]))
        assert e2c[:16].tolist() == survey_edge_to_column
        assert bp.tolist() == [2, 1, 1] and p1.tolist() == [0, 2, 3, 4]
    assert (g["hspa_packed"][8:12] == 0).all() and (g["hind"][16:24] == 0).all()


def test_format_round_trip(csr_fixture):
    g = csr_fixture
    n = int(g["num_nodes"])
    indptr, indices = oracle_np.blocked_to_csr(g["pointer1"], g["hspa_packed"], g["hind"], n)
    a = sp.csr_matrix((np.ones(len(g["indices"])), g["indices"], g["indptr"]), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    assert (indptr == a.indptr).all() and (indices == a.indices).all()


@pytest.mark.parametrize("rounding", ["none", "tf32", "fp16"])
def test_blocked_spmm_matches_torch_sparse_mm(csr_fixture, rounding):
    g = csr_fixture
    n = int(g["num_nodes"])
    ref = g["torch_sparse_mm"].astype(np.float64)
    out_c = oracle_c.spmm_blocked(g["pointer1"], g["hspa_packed"], g["hind"], n, g["feat"], rounding=rounding)
    out_np = oracle_np.spmm_blocked(g["pointer1"], g["hspa_packed"], g["hind"], n, g["feat"], rounding=rounding)
    assert np.abs(out_c - out_np).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    bound = oracle_np.forward_error_bound(g["indptr"], g["indices"], g["feat"], n,
                                          rounding="fp16" if rounding != "none" else "none")
    assert (np.abs(out_c - ref) <= bound + 1e-6 * np.abs(ref) + 1e-7).all()
    assert abs(oracle_np.calc_diff(out_c.astype(np.float64), ref)) <= 1e-5


def test_reference_test_criterion_on_reference_test_inputs():
    """tests/test_spmm_kernel.py defaults (N=8192, density 0.01, F=512, seed 20): `difference rate` 0.00 %."""
    np.random.seed(20)
    torch.manual_seed(20)
    n, f = 8192, 512
    a = sp.random(n, n, density=0.01, format="csr")
    feat = torch.randn(n, f, dtype=torch.float32)
    ref = torch_ref.spmm(a.indptr, a.indices, feat, n).numpy()
    p1, packed, hind = oracle_c.csr_preprocess(a.indptr, a.indices, n)
    for rounding, tol in (("tf32", 1e-3), ("fp16", 1e-3)):
        out = oracle_c.spmm_blocked(p1, packed, hind, n, feat.numpy(), rounding=rounding)
        assert abs(oracle_np.calc_diff(out.astype(np.float64), ref.astype(np.float64))) <= 1e-5
        assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= tol
    # with no operand rounding only the accumulation order differs
    out = oracle_c.spmm_blocked(p1, packed, hind, n, feat.numpy(), rounding="none")
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= 1e-6


def test_metrics_match_reference_python():
    gold = json.load(open(os.path.join(GOLDEN, "ref_python_goldens.json")))["metrics"]
    g = torch.Generator().manual_seed(gold["seed"])
    x = torch.randn(*gold["shape"], generator=g)
    y = x + gold["noise"] * torch.randn(*gold["shape"], generator=g)
    assert abs(float(x.double().sum()) - gold["x_sum"]) < 1e-9
    assert abs(oracle_np.calc_diff(x.double().numpy(), y.double().numpy()) - gold["calc_diff_f64"]) < 1e-12
    assert abs(oracle_np.relative_error(y.numpy(), x.numpy()) - gold["relative_error"]) < 1e-9
    from voltrix.utils import calc_diff, relative_error  # the product's own metrics, same goldens

    assert abs(float(calc_diff(x, y)) - gold["calc_diff"]) < 1e-9
    assert abs(float(calc_diff(x, y, dtype=torch.float64)) - gold["calc_diff_f64"]) < 1e-12
    assert abs(relative_error(y, x) - gold["relative_error"]) < 1e-9


def test_tf32_rna_and_fp16_rounding_bit_patterns():
    x = np.array([1.0, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, -(1.0 + 2.0 ** -11), 3.0e-5, 65504.0], np.float32)
    t = oracle_np.round_tf32_rna(x)
    assert t[0] == 1.0 and t[1] == np.float32(1.0 + 2.0 ** -10)  # tie rounds away from zero
    assert t[3] == -np.float32(1.0 + 2.0 ** -10)
    h = oracle_np.round_operand(x, "fp16")
    assert h[1] == 1.0  # fp16: ties to even
    n = x.size
    out = oracle_c.spmm_csr(np.arange(n + 1), np.arange(n), x.reshape(n, 1), n, rounding="fp16").ravel()
    assert (out == h).all()
    out = oracle_c.spmm_csr(np.arange(n + 1), np.arange(n), x.reshape(n, 1), n, rounding="tf32").ravel()
    assert (out == t).all()
