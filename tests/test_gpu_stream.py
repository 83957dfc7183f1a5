"""GPU: the stream kernel (spmm_stream_kernels.hpp, tuner schedule 6 -- the window format's kernel for short windows) against
the oracle (torch.sparse.mm on the CPU = the reference's own oracle call, tests/test_spmm.py:24-29) and against the window
kernel: stated tolerance, exact sums on integer operands, same bits as the window kernel, every width class, empty windows
with a poisoned B[0], cut windows (partial tiles + combine), bfloat16 / fp32 inputs, 64-bit addressing."""
import numpy as np
import pytest
import torch

import synth_graphs
import voltrix
from conftest import load_csr_fixture
from oracle import torch_ref
from test_gpu_spmm import _assert_close
from voltrix.jit_kernels import jit_tuner
from voltrix.jit_kernels.spmm import SCHED_STREAM, feature_hash
from voltrix.schedule import stream_tables

pytestmark = pytest.mark.gpu


def _chosen(hspa_packed, f, dtype=torch.float16):
    keys = {"feature_hash": feature_hash(hspa_packed), "embedding_dim": f, "dtype": str(dtype),
            "device": torch.cuda.get_device_name(0), "two_level": False, "weighted": False}
    return jit_tuner.tuned_point("spmm_kernel", keys)


def _stream_call(handle, n, nnz, feat, tag, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "stream")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    handle[1].hash_tag = tag
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)
    return out


@pytest.mark.parametrize("dtype,mode", [(torch.float16, "fp16"), (torch.bfloat16, "exact"), (torch.float32, "fp16-scaled"),
                                        (torch.float32, "exact")])
def test_stream_kernel_on_fixtures(cuda_device, csr_fixture, dtype, mode, monkeypatch):
    """fp32 features: the scaled-fp16 operand (VOLTRIX_FP32_MODE=fp16) and -- the default on these short-window handles -- the
    fp32 rows as they are, exact products on v_mfma_f32_16x16x4_f32 (the stream kernel's EB = 4 tiles)."""
    if dtype == torch.float32:
        monkeypatch.setenv("VOLTRIX_FP32_MODE", "fp16" if mode == "fp16-scaled" else "auto")
    g = csr_fixture
    n = int(g["num_nodes"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    feat32 = torch.from_numpy(g["feat"]).float()
    if dtype != torch.float32:
        feat32 = feat32.to(dtype).float()
    out = _stream_call(handle, n, len(g["indices"]), feat32.to(dtype).cuda(), f"stream_fixture_{n}_{dtype}_{mode}", monkeypatch)
    f = feat32.shape[1]
    exact32 = dtype == torch.float32 and mode == "exact"
    padded = (f + 3) // 4 * 4 if exact32 else (f + 7) // 8 * 8
    point = _chosen(handle[1], padded, dtype if (exact32 or dtype != torch.float32) else torch.float16)
    assert point.get("SCHED") == SCHED_STREAM and point.get("EB") == (4 if exact32 else 2)
    _assert_close(out, g["indptr"], g["indices"], feat32, n, mode)


@pytest.mark.parametrize("name,scale", [("yeast_like", 0.02), ("dd_like", 0.1), ("com_amazon_like", 0.1),
                                        ("web_berkstan_like", 0.05), ("ppi_like", 0.5)])
@pytest.mark.parametrize("f", [32, 64, 128, 200, 384])
def test_stream_kernel_is_exact_on_integers_and_matches_the_window_kernel(cuda_device, name, scale, f, monkeypatch):
    """Low-degree stand-ins of the reference's evaluation set (bench/plot.py:8): integer operands make every fp32 sum exact, so
    the stream kernel, the window kernel and the oracle must agree bit for bit."""
    indptr, indices, _ = synth_graphs.generate(name, scale=scale)
    n, nnz = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    torch.manual_seed(f)
    feat = torch.randint(-3, 4, (n, f)).half()
    out = _stream_call(handle, n, nnz, feat.cuda(), f"stream_int/{name}/{f}", monkeypatch)
    ref = torch_ref.spmm(indptr.numpy(), indices.numpy(), feat.float(), n)
    assert torch.equal(out.cpu(), ref)
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    handle[1].hash_tag = f"window_int/{name}/{f}"
    assert torch.equal(voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat.cuda()), out)


def test_stream_kernel_stated_tolerance_on_two_stand_ins(cuda_device, monkeypatch):
    """The stated fp16 tolerance on two of the round-5 stand-ins at sizes the CPU oracle finishes in
    seconds: yeasth_like (degree 2: one stage per window) and amazon0505_like (degree 12)."""
    for name, scale in (("yeasth_like", 0.05), ("amazon0505_like", 0.25)):
        indptr, indices, _ = synth_graphs.generate(name, scale=scale)
        n, nnz = indptr.numel() - 1, indices.numel()
        handle = voltrix.csr_preprocess(indptr, indices, n)
        torch.manual_seed(3)
        feat32 = torch.randn(n, 128).half().float()
        out = _stream_call(handle, n, nnz, feat32.half().cuda(), f"stream_tol/{name}", monkeypatch)
        _assert_close(out, indptr.numpy(), indices.numpy(), feat32, n, "fp16")


def test_stream_kernel_never_feeds_row_zero_of_b_to_padded_columns(cuda_device, monkeypatch):
    """Padded hind slots are 0 in the format; windows without edges own one all-zero TC block.  With
    NaN in B[0] and no edge to column 0, no output may be NaN, and rows of windows without edges must be exactly 0."""
    rng = np.random.default_rng(5)
    n = 16 * 40 + 5
    rows = []
    for r in range(n):
        w = r // 16
        if w in (0, 7, 8, 39) or rng.random() < 0.3:     # whole empty windows (first, consecutive, last full) and empty rows
            rows.append(np.zeros(0, np.int64))
        else:
            rows.append(np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 12)), replace=False)))
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    indices = np.concatenate(rows).astype(np.int32)
    handle = voltrix.csr_preprocess(torch.from_numpy(indptr), torch.from_numpy(indices), n)
    feat = torch.randn(n, 128).half()
    feat[0] = float("nan")
    out = _stream_call(handle, n, len(indices), feat.cuda(), "stream_nan_row0", monkeypatch).cpu()
    assert not torch.isnan(out).any()
    clean = feat.clone()
    clean[0] = 0
    _assert_close(out, indptr, indices, clean.float(), n, "fp16")
    for w in (0, 7, 8, 39):
        assert (out[16 * w:16 * w + 16] == 0).all()


def test_stream_tables_cut_long_windows_and_the_result_does_not_change(cuda_device, monkeypatch):
    """A graph with hub windows: the table cuts them into interleaved units (partial tiles, combine pass); integer operands:
    the same bits whatever the cut length and the run cost."""
    g = load_csr_fixture("skewed_1005")
    n, e = int(g["num_nodes"]), len(g["indices"])
    handle = voltrix.csr_preprocess(torch.from_numpy(g["indptr"]), torch.from_numpy(g["indices"]), n)
    feat = torch.randint(-2, 3, (n, 128)).half()
    ref = torch_ref.spmm(g["indptr"], g["indices"], feat.float(), n)
    from voltrix.jit_kernels import spmm as S

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "stream")
    monkeypatch.setenv("VOLTRIX_HYBRID", "0")
    for run_cost, cut in ((6, 6), (12, 3), (48, 200), (2, 1)):
        table = stream_tables(*handle, n, run_cost=run_cost, cut_stages=cut)
        assert (cut >= 200) == (table.num_cuts == 0)
        handle[1]._voltrix_stream_table = ((handle[0].data_ptr(), n), table)   # the cache handle_stream_table reads
        handle[1].hash_tag = f"stream_cuts/{run_cost}/{cut}"
        out = voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat.cuda())
        assert S.handle_stream_table(*handle, n) is table
        assert torch.equal(out.cpu(), ref), (run_cost, cut)


def test_stream_kernel_64_bit_addressing(cuda_device, monkeypatch):
    """B of 4.3 GB (> 2^32 bytes): the launcher takes the 64-bit-pointer instantiation; rows sampled against the oracle."""
    n, f = 2_100_000, 1024
    indptr, indices, _ = synth_graphs.generate("yeast_like", device="cuda", scale=n / 1710902)
    n = indptr.numel() - 1
    assert n * f * 2 > 2 ** 32
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    feat = torch.randint(-3, 4, (n, f), device="cuda", dtype=torch.int8).half()
    out = _stream_call(handle, n, indices.numel(), feat, "stream_addr64", monkeypatch)
    rows = torch.cat([torch.arange(0, 64), torch.randint(0, n, (512,)), torch.arange(n - 64, n)]).unique()
    ip, ix = indptr.cpu().long(), indices.cpu().long()
    for r in rows.tolist():
        cols = ix[ip[r]:ip[r + 1]].cuda()
        assert torch.equal(out[r], feat[cols].float().sum(0)), r


@pytest.mark.parametrize("name,scale,f", [("yeast_like", 0.02, 128), ("dd_like", 0.1, 200), ("com_amazon_like", 0.1, 36),
                                          ("yeasth_like", 0.01, 32), ("web_berkstan_like", 0.05, 64)])
def test_exact_fp32_stream_kernel(cuda_device, name, scale, f, monkeypatch):
    """fp32 features through the stream kernel's exact tiles (forced: VOLTRIX_FP32_MODE=exact): against the fp32 oracle the only
    error left is the accumulation order, deg 2^-23 (A |B|); integer operands are exact; values far outside fp16's range pass
    through untouched (no cast, no scale)."""
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "exact")
    indptr, indices, _ = synth_graphs.generate(name, scale=scale)
    n, nnz = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess(indptr, indices, n)
    torch.manual_seed(f)
    feat = torch.randn(n, f) * torch.logspace(-30, 30, n)[:, None]      # 60 decades of row scales
    out = _stream_call(handle, n, nnz, feat.cuda(), f"stream_exact/{name}/{f}", monkeypatch)
    assert _chosen(handle[1], (f + 3) // 4 * 4, torch.float32).get("EB") == 4
    _assert_close(out, indptr.numpy(), indices.numpy(), feat, n, "exact")
    ints = torch.randint(-1000, 1001, (n, f)).float()
    out = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=ints.cuda())
    assert torch.equal(out.cpu(), torch_ref.spmm(indptr.numpy(), indices.numpy(), ints, n))


def test_fp32_mode_auto_follows_the_handle(cuda_device, monkeypatch):
    """Short windows (at most six gathered rows of B per output row): the fp32 rows as they are; long windows: the scaled cast."""
    from voltrix.spmm.spmm import fp32_mode

    monkeypatch.delenv("VOLTRIX_FP32_MODE", raising=False)
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "default")
    for name, scale, want in (("yeast_like", 0.02, "exact"), ("dd_like", 0.1, "exact"), ("amazon0505_like", 0.1, "fp16"),
                              ("reddit_like", 0.02, "fp16")):
        indptr, indices, _ = synth_graphs.generate(name, scale=scale)
        n = indptr.numel() - 1
        handle = voltrix.csr_preprocess(indptr, indices, n)
        assert fp32_mode(handle[1], n) == want == fp32_mode(handle[1], n, 128), name
        # at most 32 columns: an fp32 row is one 128-byte line -- up to 16 gathered rows per output row stay exact
        assert fp32_mode(handle[1], n, 32) == ("fp16" if name == "reddit_like" else "exact"), name
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    assert fp32_mode(handle[1], n) == "fp16"


@pytest.mark.parametrize("n,deg", [(1, 0), (1, 1), (15, 2), (16, 3), (17, 1), (33, 0), (100, 5), (1000, 0)])
@pytest.mark.parametrize("f,dtype", [(8, torch.float16), (24, torch.bfloat16), (40, torch.float16), (72, torch.float32), (4, torch.float32)])
def test_stream_kernel_edge_shapes(cuda_device, n, deg, f, dtype, monkeypatch):
    """Tiny and degenerate inputs: fewer rows than a window, graphs without edges (every window owns one all-zero TC block),
    widths that leave a partial 16-column slot, every operand type -- against the oracle, exact on integer operands."""
    monkeypatch.setenv("VOLTRIX_FP32_MODE", "auto")
    rng = np.random.default_rng(n * 131 + deg)
    rows = [np.sort(rng.choice(n, size=min(deg, n), replace=False)) for _ in range(n)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    indices = (np.concatenate(rows) if deg else np.zeros(0)).astype(np.int32)
    handle = voltrix.csr_preprocess(torch.from_numpy(indptr), torch.from_numpy(indices), n)
    feat = torch.randint(-4, 5, (n, f)).to(dtype)
    out = _stream_call(handle, n, len(indices), feat.cuda(), f"stream_edge/{n}/{deg}/{f}/{dtype}", monkeypatch)
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert out.shape == (n, f) and torch.equal(out.cpu(), ref)
