#!/usr/bin/env python3
"""Dump a graph in the reference's benchmark file formats (reference bench/graph_gen.py:57-121):

    indices.csv / indptr.csv   text, one int per line (np.savetxt fmt="%d")
    feat.csv                   RAW float32 [N, F] (tofile, despite the name)
    output_base.csv            RAW float32 [N, F] = csr(ones) @ feat  (the reference computes it with cuSPARSE on the
                               GPU, :104-121; here torch.sparse.mm on the CPU -- the same oracle call)
    data.mtx                   Matrix Market pattern file (optional)

Input: ``--npz`` a TC-GNN style archive (``src_li``, ``dst_li``, ``num_nodes`` -- what the reference's datasets.zip
holds) or a scipy.sparse ``save_npz`` CSR file; or ``--synthetic NAME[:scale]`` for the seeded stand-ins of
synth_graphs.py.  Bench infrastructure, not part of the product package.
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


def write_reorder_npz(path, method):
    """NAME.npz -> NAME.reorder.npz: nodes relabelled so that position k of the order becomes node k (rows AND columns, as
    the reference's externally reordered files are)."""
    sys.path.insert(0, os.path.join(REPO, "voltrix-spmm_amd"))
    from voltrix import reorder

    indptr, indices = load_npz(path)
    n = len(indptr) - 1
    if method == "rcm":
        perm = reorder.rcm_permutation(indptr, indices, n)
    else:
        ip, ix = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
        fn = reorder.spectral_permutation if method == "spectral" else reorder.bfs_permutation
        perm = fn(ip, ix, n).cpu().numpy()
    label = np.empty(n, dtype=np.int64)
    label[perm] = np.arange(n)
    rows = np.repeat(np.arange(n), np.diff(indptr))
    out = path[:-4] + ".reorder.npz"
    np.savez(out, src_li=label[rows], dst_li=label[indices], num_nodes=n)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--npz")
    src.add_argument("--synthetic", help="NAME[:scale] from synth_graphs.CONFIGS")
    ap.add_argument("--num_feats", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=20)
    ap.add_argument("--out_dir", default=".")
    ap.add_argument("--only_dense", action="store_true", help="only feat.csv / output_base.csv (reference flag)")
    ap.add_argument("--mtx", action="store_true", help="also write data.mtx")
    ap.add_argument("--write_reorder", default=None, choices=["spectral", "bfs", "rcm"],
                    help="with --npz NAME.npz: also write NAME.reorder.npz -- the graph with its nodes relabelled (P A P^T, "
                         "TC-GNN edge-list layout) by voltrix.reorder's spectral / bfs order (GPU) or scipy's RCM (host): the "
                         "file the reference's protocol reads with --reorder (bench/graph_gen.py:42-45)")
    ap.add_argument("--reorder", action="store_true", help="with --npz NAME.npz: dump NAME.reorder.npz instead (reference flag)")
    args = ap.parse_args(argv)

    if args.npz:
        if args.write_reorder:
            write_reorder_npz(args.npz, args.write_reorder)
        indptr, indices = load_npz(args.npz[:-4] + ".reorder.npz" if args.reorder else args.npz)
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
