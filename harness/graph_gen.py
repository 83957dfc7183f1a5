#!/usr/bin/env python3
"""Dump a graph in the reference's benchmark file formats (reference bench/graph_gen.py:57-121):

    indices.csv / indptr.csv   text, one int per line (np.savetxt fmt="%d")
    feat.csv                   RAW float32 [N, F] (tofile, despite the name)
    output_base.csv            RAW float32 [N, F] = csr(ones) @ feat  (the reference computes it with cuSPARSE on the
                               GPU, :104-121; here torch.sparse.mm on the CPU -- the same oracle call)
    data.mtx                   Matrix Market pattern file (optional)

Input: ``--npz`` a TC-GNN style archive (``src_li``, ``dst_li``, ``num_nodes`` -- what the reference's datasets.zip
holds) or a scipy.sparse ``save_npz`` CSR file; ``--mtx_in`` a Matrix Market file as the SuiteSparse collection ships them
(``load_mtx``: coordinate format, pattern / real / integer / complex entries, general / symmetric / skew-symmetric / hermitian
storage expanded to both triangles; the reference only WRITES data.mtx, bench/graph_gen.py:104-121, and reads its graphs
from the TC-GNN archives -- BASELINE.json's north_star names SuiteSparse graphs, whose native format this is); or
``--synthetic NAME[:scale]`` for the seeded stand-ins of synth_graphs.py.  Bench infrastructure (SURVEY.md section 8f rank
2), not part of the product package.
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def load_npz(path):
    z = np.load(path)
    if "src_li" in z.files:  # TC-GNN layout: edge list
        n = int(z["num_nodes"])
        a = sp.coo_matrix((np.ones(len(z["src_li"]), np.float32), (z["src_li"], z["dst_li"])), shape=(n, n)).tocsr()
    else:
        a = sp.load_npz(path).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def load_mtx(path, return_values=False):
    """Matrix Market coordinate file -> CSR ``(indptr int32, indices int32[, values float32])`` of the PATTERN: rows sorted,
    duplicate entries merged (values summed), explicit zeros kept as edges (the SpMM here is binary: an entry is an edge).

    Header ``%%MatrixMarket matrix coordinate <field> <symmetry>``; ``%`` comment lines; a size line ``M N NNZ``; then NNZ
    lines ``i j [value [imag]]`` with 1-based indices.  ``symmetric`` / ``hermitian`` / ``skew-symmetric`` files store one
    triangle: the mirror entries are added (skew: negated; the diagonal is stored once).  ``array`` (dense) files are
    refused.  ``.mtx.gz`` is read through gzip.  A rectangular matrix M x N is returned as it is (indptr has M + 1 entries);
    callers that need a square adjacency check the shape."""
    import gzip
    import io

    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        header = f.readline().strip().split()
        assert len(header) >= 5 and header[0].lower() == "%%matrixmarket" and header[1].lower() == "matrix", \
            f"{path}: not a Matrix Market file"
        layout, field, symmetry = header[2].lower(), header[3].lower(), header[4].lower()
        assert layout == "coordinate", f"{path}: '{layout}' layout (dense) is not a sparse graph"
        assert field in ("pattern", "real", "integer", "double", "complex") and \
            symmetry in ("general", "symmetric", "skew-symmetric", "hermitian"), (field, symmetry)
        line = f.readline()
        while line and (line.startswith("%") or not line.strip()):
            line = f.readline()
        m, n, nnz = (int(t) for t in line.split()[:3])
        body = f.read()
    ncol = {"pattern": 2, "complex": 4}.get(field, 3)
    if nnz:
        try:    # pandas' C parser: ~10x numpy.loadtxt on SuiteSparse-sized files
            import pandas as pd

            data = pd.read_csv(io.StringIO(body), sep=r"\s+", header=None, comment="%", dtype=np.float64,
                               engine="c").to_numpy()
        except ImportError:
            data = np.loadtxt(io.StringIO(body), comments="%", ndmin=2, dtype=np.float64)
    else:
        data = np.zeros((0, ncol))
    assert data.shape[0] == nnz and data.shape[1] >= min(ncol, 2), f"{path}: {data.shape[0]} entries, header says {nnz}"
    rows = data[:, 0].astype(np.int64) - 1
    cols = data[:, 1].astype(np.int64) - 1
    vals = data[:, 2].astype(np.float32) if (field != "pattern" and data.shape[1] > 2) else np.ones(nnz, np.float32)
    assert nnz == 0 or (rows.min() >= 0 and rows.max() < m and cols.min() >= 0 and cols.max() < n), f"{path}: index out of range"
    if symmetry != "general":
        off = rows != cols
        sign = -1.0 if symmetry == "skew-symmetric" else 1.0
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, sign * vals[off]]))
    a = sp.coo_matrix((vals, (rows, cols)), shape=(m, n))
    pattern = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(m, n)).tocsr()   # explicit zeros stay edges
    pattern.sum_duplicates()
    pattern.sort_indices()
    indptr, indices = pattern.indptr.astype(np.int32), pattern.indices.astype(np.int32)
    if not return_values:
        return indptr, indices
    a = a.tocsr()
    a.sum_duplicates()
    a.sort_indices()
    # values aligned with the pattern (an entry whose duplicates cancel to 0 is still an edge, with value 0)
    dense_vals = np.zeros(len(indices), np.float32)
    prow = np.repeat(np.arange(m), np.diff(indptr))
    dense_vals[:] = np.asarray(a[prow, indices]).ravel()
    return indptr, indices, dense_vals


def load_graph(path):
    """Dispatch on the extension: ``.npz`` (TC-GNN archive or scipy CSR) or ``.mtx`` / ``.mtx.gz`` (Matrix Market)."""
    if path.endswith(".mtx") or path.endswith(".mtx.gz"):
        indptr, indices = load_mtx(path)
        assert len(indptr) - 1 >= (int(indices.max()) + 1 if len(indices) else 0), f"{path}: not a square adjacency"
        return indptr, indices
    return load_npz(path)


def write_reorder_npz(path, method):
    """NAME.npz -> NAME.reorder.npz: nodes relabelled so that position k of the order becomes node k (rows AND columns, as
    the reference's externally reordered files are).  The relabelled CSR is the library's own (``voltrix.reorder.relabel_csr``,
    what ``csr_preprocess_reordered(..., relabel=True)`` runs on); ``method`` "auto" = that call's default, candidates judged
    with B's address locality, the caller's order kept when nothing pays."""
    sys.path.insert(0, os.path.join(REPO, "voltrix-spmm_amd"))
    from voltrix import reorder

    indptr, indices = load_npz(path)
    n = len(indptr) - 1
    ip, ix = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    if method == "rcm":
        perm = torch.from_numpy(np.ascontiguousarray(reorder.rcm_permutation(indptr, indices, n))).cuda().long()
    elif method == "auto":
        perm, _ = reorder.auto_permutation(ip, ix, n, relabel=True)
        perm = torch.arange(n, device="cuda") if perm is None else perm
    else:
        perm = (reorder.spectral_permutation if method == "spectral" else reorder.bfs_permutation)(ip, ix, n)
    r_indptr, r_indices = reorder.relabel_csr(ip, ix, n, perm)
    rows = np.repeat(np.arange(n), np.diff(r_indptr.cpu().numpy()))
    out = path[:-4] + ".reorder.npz"
    np.savez(out, src_li=rows, dst_li=r_indices.cpu().numpy().astype(np.int64), num_nodes=n)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--npz")
    src.add_argument("--mtx_in", help="a Matrix Market file (SuiteSparse collection format; .mtx or .mtx.gz)")
    src.add_argument("--synthetic", help="NAME[:scale] from synth_graphs.CONFIGS")
    ap.add_argument("--num_feats", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=20)
    ap.add_argument("--out_dir", default=".")
    ap.add_argument("--only_dense", action="store_true", help="only feat.csv / output_base.csv (reference flag)")
    ap.add_argument("--mtx", action="store_true", help="also write data.mtx")
    ap.add_argument("--write_reorder", default=None, choices=["auto", "spectral", "bfs", "rcm"],
                    help="with --npz NAME.npz: also write NAME.reorder.npz -- the graph with its nodes relabelled (P A P^T, "
                         "TC-GNN edge-list layout) by voltrix.reorder's spectral / bfs order (GPU) or scipy's RCM (host): the "
                         "file the reference's protocol reads with --reorder (bench/graph_gen.py:42-45)")
    ap.add_argument("--reorder", action="store_true", help="with --npz NAME.npz: dump NAME.reorder.npz instead (reference flag)")
    ap.add_argument("--no_dump", action="store_true", help="with --write_reorder: only write NAME.reorder.npz, no CSV / raw files")
    args = ap.parse_args(argv)

    if args.npz:
        if args.write_reorder:
            write_reorder_npz(args.npz, args.write_reorder)
            if args.no_dump:
                return
        indptr, indices = load_npz(args.npz[:-4] + ".reorder.npz" if args.reorder else args.npz)
    elif args.mtx_in:
        indptr, indices = load_graph(args.mtx_in)
    else:
        import synth_graphs

        name, _, scale = args.synthetic.partition(":")
        ip, ix, _ = synth_graphs.generate(name, scale=float(scale or 1.0))
        indptr, indices = ip.numpy(), ix.numpy()
    n = len(indptr) - 1
    os.makedirs(args.out_dir, exist_ok=True)
    out = lambda f: os.path.join(args.out_dir, f)  # noqa: E731
    if not args.only_dense:
        np.savetxt(out("indices.csv"), indices, delimiter=",", fmt="%d")
        np.savetxt(out("indptr.csv"), indptr, delimiter=",", fmt="%d")
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    feat = torch.randn(n, args.num_feats, dtype=torch.float32)
    feat.numpy().tofile(out("feat.csv"))
    csr = torch.sparse_csr_tensor(torch.from_numpy(indptr), torch.from_numpy(indices),
                                  torch.ones(len(indices), dtype=torch.float32), size=(n, n))
    (csr @ feat).numpy().astype(np.float32).tofile(out("output_base.csv"))
    if args.mtx:
        from scipy.io import mmwrite

        mmwrite(out("data.mtx"), sp.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n)).tocoo(),
                field="pattern")
    print(f"N={n} nnz={len(indices)} F={args.num_feats} -> {os.path.abspath(args.out_dir)}")


if __name__ == "__main__":
    main()
