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
                          "--feature_dims", "64", "--reorder_method", "bfs", "--iters", "3", "--methods", "hipSPARSE,rocSPARSE-best,Voltrix",
                          "--output_file", str(tmp_path / "results.csv")],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [r.split(",") for r in open(tmp_path / "results.csv").read().strip().split("\n")]
    assert rows[0] == ["Method", "Dataset", "FeatDim", "Reorder", "Time (ms)"]
    assert [r[:4] for r in rows[1:]] == [["hipSPARSE", "reddit_like", "64", "N"], ["rocSPARSE-best", "reddit_like", "64", "N"],
                                         ["Voltrix", "reddit_like", "64", "N"], ["Voltrix", "reddit_like", "64", "Y"]]
    assert all(0 < float(r[4]) < 100 for r in rows[1:]), rows


def _write_mtx(path, header, entries, m, n, comments=("% a comment", "%")):
    with open(path, "w") as f:
        f.write(header + "\n")
        for c in comments:
            f.write(c + "\n")
        f.write(f"{m} {n} {len(entries)}\n")
        for e in entries:
            f.write(" ".join(str(x) for x in e) + "\n")


def test_matrix_market_reader(tmp_path):
    """harness/graph_gen.py::load_mtx against scipy.io.mmread on the forms the SuiteSparse collection uses: pattern general,
    real symmetric (one triangle stored, diagonal once), integer skew-symmetric, duplicates, comment lines, gzip, and the
    file graph_gen.py itself writes (data.mtx)."""
    import gzip

    import graph_gen
    from scipy.io import mmread

    rng = np.random.default_rng(3)
    n = 37
    # 1. pattern general, with a duplicate entry and an empty row
    ent = sorted({(int(i), int(j)) for i, j in rng.integers(1, n + 1, (150, 2)) if i != 5})
    ent.append(ent[0])
    _write_mtx(tmp_path / "p.mtx", "%%MatrixMarket matrix coordinate pattern general", ent, n, n)
    ip, ix = graph_gen.load_mtx(str(tmp_path / "p.mtx"))
    ref = sp.coo_matrix((np.ones(len(ent)), ([e[0] - 1 for e in ent], [e[1] - 1 for e in ent])), shape=(n, n)).tocsr()
    ref.sum_duplicates()
    ref.sort_indices()
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices) and ip[5] == ip[4]       # row 5 (1-based) is empty
    # 2. real symmetric: lower triangle + diagonal, expanded to both triangles; values follow
    low = sorted({(int(max(i, j)), int(min(i, j))) for i, j in rng.integers(1, n + 1, (120, 2))})
    ent = [(i, j, round(float(v), 3)) for (i, j), v in zip(low, rng.normal(size=len(low)))]
    _write_mtx(tmp_path / "s.mtx", "%%MatrixMarket matrix coordinate real symmetric", ent, n, n)
    ip, ix, vals = graph_gen.load_mtx(str(tmp_path / "s.mtx"), return_values=True)
    ref = mmread(str(tmp_path / "s.mtx")).tocsr()
    ref.sort_indices()
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices) and np.allclose(vals, ref.data, atol=1e-6)
    dense = np.zeros((n, n), bool)
    dense[np.repeat(np.arange(n), np.diff(ip)), ix] = True
    assert (dense == dense.T).all()
    # 3. integer skew-symmetric (no diagonal): mirrored with the sign flipped; the PATTERN is symmetric
    ent = [(i, j, int(v)) for (i, j), v in zip([e for e in low if e[0] != e[1]], rng.integers(1, 9, len(low)))]
    _write_mtx(tmp_path / "k.mtx", "%%MatrixMarket matrix coordinate integer skew-symmetric", ent, n, n)
    ip, ix, vals = graph_gen.load_mtx(str(tmp_path / "k.mtx"), return_values=True)
    ref = mmread(str(tmp_path / "k.mtx")).tocsr()
    ref.sort_indices()
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices) and np.allclose(vals, ref.data)
    # 4. gzip, and graph_gen's own data.mtx round trip through load_graph
    with open(tmp_path / "p.mtx", "rb") as src, gzip.open(tmp_path / "p2.mtx.gz", "wb") as dst:
        dst.write(src.read())
    ip2, ix2 = graph_gen.load_graph(str(tmp_path / "p2.mtx.gz"))
    ip1, ix1 = graph_gen.load_mtx(str(tmp_path / "p.mtx"))
    assert np.array_equal(ip1, ip2) and np.array_equal(ix1, ix2)
    g, nn = _make(tmp_path)
    ip3, ix3 = graph_gen.load_graph(str(tmp_path / "data.mtx"))
    assert np.array_equal(ip3, g["indptr"]) and np.array_equal(ix3, g["indices"])
    # 5. the --mtx_in source of graph_gen.py writes the reference's files from a SuiteSparse-format input
    graph_gen.main(["--mtx_in", str(tmp_path / "s.mtx"), "--num_feats", "8", "--out_dir", str(tmp_path / "from_mtx")])
    assert np.array_equal(np.loadtxt(tmp_path / "from_mtx" / "indptr.csv", delimiter=",", dtype=np.int32), ref_indptr(tmp_path))
    # 6. dense ('array') files and malformed headers are refused loudly
    (tmp_path / "d.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(AssertionError):
        graph_gen.load_mtx(str(tmp_path / "d.mtx"))


def ref_indptr(tmp_path):
    from scipy.io import mmread

    return mmread(str(tmp_path / "s.mtx")).tocsr().indptr.astype(np.int32)


def test_experiment_scripts_only_call_entry_points_that_exist():
    """harness/experiments/*.py are run by hand on the GPU box and rot silently: every ``capi.X`` / ``schedule.X`` / ``hybrid.X`` /
    ``voltrix.X`` attribute they name must exist in the package as it stands (five scripts called removed entry points)."""
    import ast
    import glob
    import importlib

    import voltrix

    modules = {"capi": importlib.import_module("voltrix.capi"), "schedule": importlib.import_module("voltrix.schedule"),
               "hybrid": importlib.import_module("voltrix.hybrid"), "reorder": importlib.import_module("voltrix.reorder"),
               "voltrix": voltrix}
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "harness", "experiments")
    scripts = sorted(glob.glob(os.path.join(root, "*.py")) + glob.glob(os.path.join(root, "*", "*.py")))
    assert scripts
    missing = []
    for path in scripts:
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)        # also: the script is valid Python
        aliases = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module == "voltrix":
                for a in node.names:
                    if a.name in modules:
                        aliases[a.asname or a.name] = a.name
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name == "voltrix":
                        aliases[a.asname or "voltrix"] = "voltrix"
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in aliases:
                mod = modules[aliases[node.value.id]]
                if not hasattr(mod, node.attr):
                    missing.append(f"{os.path.relpath(path, root)}: {node.value.id}.{node.attr}")
    assert not missing, missing


def test_rocsparse_baseline_library_builds_and_exports_its_entry_points():
    """harness/bm_rocsparse.cpp (round 6: rocSPARSE's four CSR SpMM algorithms + the plain CSR row-gather kernel; plain HIP + rocSPARSE, no
    torch) is compiled ahead of time by build() / `make -C harness`; no GPU call here."""
    import ctypes
    import subprocess as sp

    sp.check_call(["make", "-s", "-C", os.path.join(REPO, "harness"), "libbm_rocsparse.so"])
    lib = ctypes.CDLL(os.path.join(REPO, "harness", "libbm_rocsparse.so"))
    assert hasattr(lib, "bm_rocsparse_spmm") and hasattr(lib, "bm_csr_row_gather")
    sys.path.insert(0, REPO)
    from harness import bm_rocsparse

    assert set(bm_rocsparse.ALGORITHMS.values()) == {1, 4, 5, 9}          # csr, csr_row_split, csr_nnz_split (= merge), csr_merge_path
    assert bm_rocsparse.best({"rocsparse_csr": 2.0, "rocsparse_csr_row_split": None, "csr_row_gather_u4": 1.0}) == ("rocsparse_csr", 2.0)
    assert bm_rocsparse.best({"rocsparse_csr": None}) == (None, None)
