"""The reference's benchmark file formats (bench/graph_gen.py, bench/bm_voltrix.py): CPU round trip of the files, GPU
run of the timing script on them."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import REPO, load_csr_fixture

sys.path.insert(0, os.path.join(REPO, "harness"))


def _make(tmp_path, feats=24):
    import graph_gen

    g = load_csr_fixture("skewed_1005")
    n = int(g["num_nodes"])
    a = sp.csr_matrix((np.ones(len(g["indices"]), np.float32), g["indices"], g["indptr"]), shape=(n, n))
    sp.save_npz(tmp_path / "g.npz", a)
    graph_gen.main(["--npz", str(tmp_path / "g.npz"), "--num_feats", str(feats), "--out_dir", str(tmp_path), "--mtx"])
    return g, n


def test_graph_gen_file_formats(tmp_path):
    g, n = _make(tmp_path)
    assert np.array_equal(np.loadtxt(tmp_path / "indptr.csv", delimiter=",", dtype=np.int32), g["indptr"])
    assert np.array_equal(np.loadtxt(tmp_path / "indices.csv", delimiter=",", dtype=np.int32), g["indices"])
    feat = np.fromfile(tmp_path / "feat.csv", dtype=np.float32).reshape(n, 24)
    base = np.fromfile(tmp_path / "output_base.csv", dtype=np.float32).reshape(n, 24)
    a = sp.csr_matrix((np.ones(len(g["indices"])), g["indices"], g["indptr"]), shape=(n, n))
    assert np.allclose(base, a @ feat, rtol=1e-5, atol=1e-5)
    from scipy.io import mmread

    assert (mmread(tmp_path / "data.mtx").tocsr() != a).nnz == 0
    # TC-GNN style archive (src_li / dst_li / num_nodes) loads to the same CSR
    import graph_gen

    coo = a.tocoo()
    np.savez(tmp_path / "tc.npz", src_li=coo.row, dst_li=coo.col, num_nodes=n)
    ip, ix = graph_gen.load_npz(str(tmp_path / "tc.npz"))
    assert np.array_equal(ip, g["indptr"]) and np.array_equal(ix, g["indices"])


@pytest.mark.gpu
def test_bm_voltrix_on_generated_files(tmp_path, cuda_device):
    _make(tmp_path, feats=64)
    env = dict(os.environ, VOLTRIX_TUNE_SPACE="none")
    out = subprocess.run([sys.executable, os.path.join(REPO, "harness", "bm_voltrix.py"), "--dir", str(tmp_path),
                          "--dataset", "skewed", "--csv", str(tmp_path / "results.csv"), "--iters", "3"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "difference rate: 0.000%" in out.stdout and "[Voltrix] time:" in out.stdout
    rows = open(tmp_path / "results.csv").read().strip().split("\n")
    assert rows[0] == "Method,Dataset,FeatDim,Reorder,Time (ms)" and rows[1].startswith("voltrix,skewed,64,False,")


@pytest.mark.gpu
def test_bm_voltrix_consumes_a_reorder_npz(tmp_path, cuda_device):
    """The reference's reorder protocol (bench/graph_gen.py:42-45, bench_all.py:120-129): NAME.reorder.npz beside NAME.npz.
    graph_gen.py --write_reorder produces it from voltrix.reorder's spectral order, bm_voltrix.py --npz NAME.npz --reorder
    runs on it and writes the results.csv row with Reorder=True."""
    import graph_gen
    import synth_graphs

    ip, ix, _ = synth_graphs.generate("reddit_shuffled", scale=0.02)
    n = ip.numel() - 1
    a = sp.csr_matrix((np.ones(ix.numel(), np.float32), ix.numpy(), ip.numpy()), shape=(n, n))
    coo = a.tocoo()
    np.savez(tmp_path / "g.npz", src_li=coo.row, dst_li=coo.col, num_nodes=n)
    out = graph_gen.write_reorder_npz(str(tmp_path / "g.npz"), "spectral")
    assert out == str(tmp_path / "g.reorder.npz")
    rip, rix = graph_gen.load_npz(out)
    assert len(rix) == ix.numel() and sorted(np.diff(rip).tolist()) == sorted(np.diff(ip.numpy()).tolist())   # a relabelling
    env = dict(os.environ, VOLTRIX_TUNE_SPACE="none")
    for flag, mark in (([], "False"), (["--reorder"], "True")):
        run = subprocess.run([sys.executable, os.path.join(REPO, "harness", "bm_voltrix.py"), "--npz", str(tmp_path / "g.npz"),
                              "--dataset", "g", "--num_feats", "64", "--csv", str(tmp_path / "results.csv"), "--iters", "3",
                              *flag], capture_output=True, text=True, env=env, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        assert "difference rate: 0.000%" in run.stdout and "[Voltrix] time:" in run.stdout
    rows = open(tmp_path / "results.csv").read().strip().split("\n")
    assert rows[1].startswith("voltrix,g,64,False,") and rows[2].startswith("voltrix,g,64,True,")


@pytest.mark.gpu
def test_bench_all_sweeps_methods_and_writes_the_reference_csv(tmp_path, cuda_device):
    """harness/bench_all.py (reference bench/bench_all.py:62-172): dataset folder x feature widths x methods, every method in
    its own process, time lines scraped, rows ``Method,Dataset,FeatDim,Reorder,Time (ms)`` with Reorder N / Y; the missing
    <name>.reorder.npz is produced by the breadth-first order on the device."""
    env = dict(os.environ, VOLTRIX_TUNE_SPACE="none")
    out = subprocess.run([sys.executable, os.path.join(REPO, "harness", "bench_all.py"), "--synthetic", "reddit_like:0.02",
                          "--feature_dims", "64", "--reorder_method", "bfs", "--iters", "3",
                          "--output_file", str(tmp_path / "results.csv")],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [r.split(",") for r in open(tmp_path / "results.csv").read().strip().split("\n")]
    assert rows[0] == ["Method", "Dataset", "FeatDim", "Reorder", "Time (ms)"]
    assert [r[:4] for r in rows[1:]] == [["hipSPARSE", "reddit_like", "64", "N"], ["Voltrix", "reddit_like", "64", "N"],
                                         ["Voltrix", "reddit_like", "64", "Y"]]
    assert all(0 < float(r[4]) < 100 for r in rows[1:]), rows
