#!/usr/bin/env python3
"""Time ``voltrix.spmm`` on the files written by graph_gen.py (reference bench/bm_voltrix.py:12-37): prints the
reference's two lines (``difference rate`` vs output_base.csv and ``[Voltrix] time: X ms``, the line
bench/bench_all.py:26,143-144 scrapes) plus GFLOP/s and algorithmic GB/s, and can append the reference's results.csv
row ``Method,Dataset,FeatDim,Reorder,Time (ms)`` (bench_all.py:75)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "voltrix-spmm_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import voltrix  # noqa: E402
from voltrix.utils import GPU_bench, calc_diff  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=".")
    ap.add_argument("--dataset", default="dataset")
    ap.add_argument("--fp16", action="store_true", help="hand the features over as fp16 (default: fp32 like the reference)")
    ap.add_argument("--csv", default=None, help="append a results.csv row")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--npz", default=None,
                    help="read the graph from <name>.npz (TC-GNN edge-list archive or a scipy CSR) or <name>.mtx[.gz] (Matrix "
                         "Market, the SuiteSparse collection's format) instead of indices.csv / indptr.csv, as graph_gen.py reads "
                         "them; features are then drawn here (--num_feats, --seed) and the baseline is computed with "
                         "torch.sparse.mm on the CPU")
    ap.add_argument("--reorder", action="store_true",
                    help="with --npz NAME.npz: run on NAME.reorder.npz, the externally reordered file the reference's protocol "
                         "expects beside it (bench/graph_gen.py:42-45; written by `graph_gen.py --write_reorder`) -- or, when "
                         "there is no such file, on the library's own symmetric reorder of NAME.npz "
                         "(csr_preprocess_reordered(..., relabel=True)) -- and mark the "
                         "results.csv row Reorder=True")
    ap.add_argument("--num_feats", type=int, default=128)
    ap.add_argument("--seed", type=int, default=20)
    args = ap.parse_args(argv)
    f = lambda name: os.path.join(args.dir, name)  # noqa: E731

    o_base = None
    if args.npz:
        from harness.graph_gen import load_graph

        path = args.npz[:-4] + ".reorder.npz" if args.reorder else args.npz
        in_library = args.reorder and not os.path.exists(path)   # no reordered twin on disk: the library relabels (below)
        ip, ix = load_graph(args.npz if in_library else path)
        indptr, indices = torch.from_numpy(ip), torch.from_numpy(ix)
        n = indptr.numel() - 1
        torch.manual_seed(args.seed)
        weight32 = torch.randn(n, args.num_feats, dtype=torch.float32)
        o_base = torch.sparse_csr_tensor(indptr, indices, torch.ones(indices.numel()), size=(n, n)) @ weight32
        weight = weight32.cuda()
    else:
        assert not args.reorder, "--reorder names a <name>.reorder.npz: use it with --npz"
        indices = torch.tensor(np.loadtxt(f("indices.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32)
        indptr = torch.tensor(np.loadtxt(f("indptr.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32)
        n = indptr.numel() - 1
        weight = torch.tensor(np.fromfile(f("feat.csv"), dtype=np.float32)).cuda().view(n, -1)
    if args.fp16:
        weight = weight.half()
    if args.npz and in_library:
        # the library's symmetric reorder (P A P^T, candidates judged by voltrix.reorder.auto_permutation): B goes in in the
        # new order, C comes out in it -- the permutations are outside the timed call, as when the reference reads a reordered
        # file -- and the result is put back only for the comparison with the baseline
        rh = voltrix.csr_preprocess_reordered(indptr, indices, n, method="auto", relabel=True)
        rh.hspa_packed.hash_tag = f"{args.dataset}.reorder_{n}_{indices.numel()}"
        weight_in = voltrix.permute_features(rh, weight)

        def spmm():
            return voltrix.spmm_reordered(rh, weight_in)

        o = voltrix.unpermute_output(rh, spmm()).detach().cpu()
        print(f"[Voltrix] reorder: {rh.method}")
    else:
        blk_ofs, hspa_packed, hind = voltrix.csr_preprocess(indptr, indices, n)
        hspa_packed.hash_tag = f"{args.dataset}{'.reorder' if args.reorder else ''}_{n}_{indices.numel()}"

        def spmm():
            return voltrix.spmm(blk_ofs, hspa_packed, hind, num_nodes=n, num_edges=indices.numel(), feat=weight)

        o = spmm().detach().cpu()
    if o_base is None:
        o_base = torch.tensor(np.fromfile(f("output_base.csv"), dtype=np.float32).reshape(*o.shape))
    print(f"difference rate: {calc_diff(o, o_base) * 100:.3f}%")
    ms = GPU_bench(spmm, iters=args.iters, warmup=10, kernel_name="spmm")
    print(f"[Voltrix] time: {ms:.4f} ms")
    nnz, feats = indices.numel(), weight.shape[1]
    alg = 4 * (nnz + n + 1) + n * feats * (weight.element_size() + 4)
    print(f"[Voltrix] {2 * nnz * feats / ms / 1e6:.1f} GFLOP/s, algorithmic {alg / ms / 1e6:.1f} GB/s")
    if args.csv:
        new = not os.path.exists(args.csv)
        with open(args.csv, "a") as out:
            if new:
                out.write("Method,Dataset,FeatDim,Reorder,Time (ms)\n")
            out.write(f"voltrix,{args.dataset},{feats},{bool(args.reorder)},{ms:.4f}\n")


if __name__ == "__main__":
    main()
