"""Round 6: the symmetric reorder on LABEL-SHUFFLED copies of the reference's 12 evaluation graphs
(bench_all.py:120-149 times Voltrix on reordered files of all of them).  Per graph, F = 128 fp16 unless stated:
  natural_ms            the stand-in as generated (the order a reordered file has)
  shuffled_ms           P A P^T for a seeded random P (what a dataset looks like before anybody reordered it), no reorder
  reordered_ms          csr_preprocess_reordered(shuffled, method=..., relabel=True), B and C in the new order
plus what `auto` picked, its wall time and -- because the generating order is known -- the share of edges within +- 4096 labels
before and after, and the rank correlation of the new order with the generating one.
    python harness/experiments/exp_reorder_eval.py [graphs] [feat] [methods]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import reorder  # noqa: E402


def time_ms(fn, iters=20, rounds=3):
    for _ in range(5):
        fn()
    best = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        best.append(s.elapsed_time(e) / iters)
    return sorted(best)[len(best) // 2]


def step_of(indptr, indices, n, feat, tag):
    h = voltrix.csr_preprocess_device(indptr, indices, n)
    h[1].hash_tag = tag
    e = indices.numel()
    return time_ms(lambda: voltrix.spmm(*h, num_nodes=n, num_edges=e, feat=feat))


def main():
    graphs = (sys.argv[1] if len(sys.argv) > 1 else ",".join(synth_graphs.EVALUATION_SET.values())).split(",")
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    methods = (sys.argv[3] if len(sys.argv) > 3 else "auto").split(",")
    dev = torch.device("cuda", 0)
    for graph in graphs:
        indptr, indices, _ = synth_graphs.generate(graph, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        torch.manual_seed(7)
        feat = torch.randn(n, feat_dim, device=dev).half()
        line = {"graph": graph, "N": n, "nnz": e, "F": feat_dim,
                "natural_ms": step_of(indptr, indices, n, feat, f"exp_reorder_eval/{graph}/natural"),
                "natural_local": reorder.local_fraction(indptr, indices, n)}
        s_indptr, s_indices, label = synth_graphs.shuffle_labels(indptr, indices, 1000 + len(graph))
        del indptr, indices
        line["shuffled_ms"] = step_of(s_indptr, s_indices, n, feat, f"exp_reorder_eval/{graph}/shuffled")
        line["shuffled_local"] = reorder.local_fraction(s_indptr, s_indices, n)
        true_pos = torch.empty(n, dtype=torch.int64, device=dev)     # shuffled id -> generating position
        true_pos[label] = torch.arange(n, device=dev)
        for method in methods:
            info = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                h = voltrix.csr_preprocess_reordered(s_indptr, s_indices, n, method=method, relabel=True, info=info)
            except Exception as exc:  # noqa: BLE001
                line[method] = {"error": repr(exc)[:300]}
                continue
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            fin = voltrix.permute_features(h, feat)
            new_label = torch.empty(n, dtype=torch.int64, device=dev)
            new_label[h.perm] = torch.arange(n, device=dev)
            a, b = true_pos[h.perm].double(), torch.arange(n, device=dev).double()
            corr = float(torch.corrcoef(torch.stack([a, b]))[0, 1])
            rep = info.get("report") or {}
            line[method] = {"picked": h.method, "wall_ms": round(wall, 1),
                            "reordered_ms": time_ms(lambda: voltrix.spmm_reordered(h, fin, hash_tag=f"exp_reorder_eval/{graph}/{method}")),
                            "local": reorder.local_fraction(s_indptr, s_indices, n, new_label), "corr_with_generating_order": round(corr, 4),
                            "two_level": voltrix.two_level_of(h.hspa_packed) is not None,
                            "estimates": {k: round(v.get("estimated_ms", 0.0), 4) for k, v in rep.items()},
                            "candidates": {k: {kk: vv for kk, vv in v.items() if kk in ("tc_blocks", "local_fraction", "accepted", "two_level", "shared_fraction")}
                                           for k, v in rep.items()}}
            r = line[method]
            r["vs_natural"] = round(r["reordered_ms"] / line["natural_ms"], 3)
            r["vs_shuffled"] = round(r["reordered_ms"] / line["shuffled_ms"], 3)
            del h, fin
        print(json.dumps(line), flush=True)
        del s_indptr, s_indices, feat
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
