"""How far is the spectral order of a label-shuffled band graph from the order the graph was generated in?  The shuffle is
seeded (synth_graphs.shuffle_labels), so the true position of every row is known: per decile of the recovered order the
median / 90 % displacement from a monotone fit, the mean original distance of adjacent rows, and whether the order is folded
(original index not monotone along the recovered order).
    python harness/experiments/exp_spectral_error.py [--graph reddit_like] [--refine 0,4]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
from voltrix import reorder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="reddit_like")
    ap.add_argument("--seed", type=int, default=101)
    ap.add_argument("--refine", default="0,4")
    ap.add_argument("--vectors", type=int, default=32)
    ap.add_argument("--iterations", type=int, default=16)
    ap.add_argument("--vote_rounds", default="2")
    ap.add_argument("--min_one_dimensional", type=float, default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    indptr, indices, _ = synth_graphs.generate(args.graph, device=dev)
    n = indptr.numel() - 1
    s_indptr, s_indices, label = synth_graphs.shuffle_labels(indptr, indices, args.seed)     # label[old] = new
    orig_of_new = torch.empty(n, dtype=torch.int64, device=dev)
    orig_of_new[label] = torch.arange(n, device=dev)
    if args.min_one_dimensional is not None:
        reorder.UNFOLD_MIN_ONE_DIMENSIONAL = args.min_one_dimensional
    for refine, rounds in [(int(x), int(v)) for x in args.refine.split(",") for v in args.vote_rounds.split(",")]:
        perm, info = reorder.spectral_permutation(s_indptr, s_indices, n, refine=refine, vectors=args.vectors,
                                                  iterations=args.iterations, return_info=True, vote_rounds=rounds)
        true_pos = orig_of_new[perm].double()            # original index of the row at recovered position k
        k = torch.arange(n, device=dev, dtype=torch.float64)
        corr = float(torch.corrcoef(torch.stack([k, true_pos]))[0, 1])
        if corr < 0:
            true_pos = (n - 1) - true_pos
        # monotone fit: running median over 2049 neighbours of the recovered order
        w = 2049
        pad = torch.nn.functional.pad(true_pos[None, None, :], (w // 2, w // 2), mode="replicate")[0, 0]
        fit = pad.unfold(0, w, 1).median(dim=1).values
        err = (true_pos - fit).abs()
        adj = (true_pos[1:] - true_pos[:-1]).abs()
        line = {"graph": args.graph, "refine": refine, "vote_rounds": rounds, "eigenvalues": [round(v, 5) for v in info["eigenvalues"]],
                "corr": round(abs(corr), 5), "phase_ms": info.get("phase_ms"), "unfolded": info.get("unfolded"), "extra_rounds": info.get("extra_rounds"), "fit_monotone_frac": round(float((fit[1:] >= fit[:-1]).double().mean()), 4),
                "fit_range": [round(float(fit.min())), round(float(fit.max()))]}
        deciles = []
        for d in range(10):
            a, b = d * n // 10, (d + 1) * n // 10
            e = err[a:b]
            deciles.append({"median_err": round(float(e.median())), "p90_err": round(float(e.quantile(0.9))),
                            "mean_adjacent": round(float(adj[a:min(b, n - 1)].mean())),
                            "fit_span": round(float(fit[b - 1] - fit[a]))})
        line["deciles"] = deciles
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
