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
    args = ap.parse_args(argv)
    f = lambda name: os.path.join(args.dir, name)  # noqa: E731

    indices = torch.tensor(np.loadtxt(f("indices.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32)
    indptr = torch.tensor(np.loadtxt(f("indptr.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32)
    n = indptr.numel() - 1
    weight = torch.tensor(np.fromfile(f("feat.csv"), dtype=np.float32)).cuda().view(n, -1)
    if args.fp16:
        weight = weight.half()
    blk_ofs, hspa_packed, hind = voltrix.csr_preprocess(indptr, indices, n)
    hspa_packed.hash_tag = f"{args.dataset}_{n}_{indices.numel()}"

    def spmm():
        return voltrix.spmm(blk_ofs, hspa_packed, hind, num_nodes=n, num_edges=indices.numel(), feat=weight)

    o = spmm().detach().cpu()
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
            out.write(f"voltrix,{args.dataset},{feats},False,{ms:.4f}\n")


if __name__ == "__main__":
    main()
