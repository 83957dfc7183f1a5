#!/usr/bin/env python3
"""Time the vendor library's CSR SpMM on the files written by graph_gen.py -- the role of the reference's
bench/bm_sparse.py:6-52 (cuSPARSE through ``torch.sparse``), here hipSPARSE through ``torch.sparse`` on ROCm: prints
``True`` / ``False`` (allclose to output_base.csv, atol 1e-1 like the reference) and ``[hipSPARSE] Elapsed time: X ms``,
the line harness/bench_all.py scrapes (reference: bench/bench_all.py:27)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=".")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--npz", default=None, help="read the graph from <name>.npz / <name>.mtx instead of the CSV dump (features "
                                                "drawn here: --num_feats, --seed; no output_base.csv check)")
    ap.add_argument("--num_feats", type=int, default=128)
    ap.add_argument("--seed", type=int, default=20)
    args = ap.parse_args(argv)
    f = lambda name: os.path.join(args.dir, name)  # noqa: E731
    if args.npz:
        sys.path.insert(0, REPO)
        from harness.graph_gen import load_graph

        ip, ix = load_graph(args.npz)
        offsets, indices = torch.from_numpy(ip).cuda(), torch.from_numpy(ix).cuda()
        n = offsets.numel() - 1
        torch.manual_seed(args.seed)
        weight = torch.randn(n, args.num_feats, dtype=torch.float32).cuda()
    else:
        indices = torch.tensor(np.loadtxt(f("indices.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32).cuda()
        offsets = torch.tensor(np.loadtxt(f("indptr.csv"), delimiter=",", dtype=np.int32), dtype=torch.int32).cuda()
        n = offsets.numel() - 1
        weight = torch.tensor(np.fromfile(f("feat.csv"), dtype=np.float32)).cuda().view(n, -1)
    csr = torch.sparse_csr_tensor(offsets, indices, values=torch.ones_like(indices).float(), size=(n, n)).cuda()
    for _ in range(10):
        out = csr @ weight
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(args.iters):
        out = csr @ weight
    end.record()
    torch.cuda.synchronize()
    if not args.npz:
        base = np.fromfile(f("output_base.csv"), dtype=np.float32).reshape(*out.shape)
        print(bool(np.allclose(out.cpu().numpy(), base, atol=1e-1)))
    ms = start.elapsed_time(end) / args.iters
    print(f"[hipSPARSE] Elapsed time: {ms:.4f} ms")
    nnz, feats = indices.numel(), weight.shape[1]
    print(f"[hipSPARSE] {2 * nnz * feats / ms / 1e6:.1f} GFLOP/s")


if __name__ == "__main__":
    main()
