"""Round 5: the symmetric reorder (csr_preprocess_reordered(..., relabel=True): P A P^T, B and C in the new order) at BASELINE
scale -- what auto's statistics say with the locality term, what is picked, the reorder's cost and the operator's step against
the un-reordered graph and against the row-only reorder (reddit_shuffled <= 1.42 ms, products_shuffled
<= 4.6 ms at F = 128 with the permutation outside the step).
    python harness/experiments/exp_reorder_relabel.py reddit_shuffled,products_shuffled,reddit_like,products_like [feat]"""
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


def time_ms(fn, iters=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    graphs = (sys.argv[1] if len(sys.argv) > 1 else "reddit_shuffled,products_shuffled,reddit_like,products_like").split(",")
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    methods = (sys.argv[3] if len(sys.argv) > 3 else "auto").split(",")
    dev = torch.device("cuda", 0)
    for graph in graphs:
        indptr, indices, _ = synth_graphs.generate(graph, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, feat_dim, device=dev).half()
        plain = voltrix.csr_preprocess_device(indptr, indices, n)
        plain[1].hash_tag = f"exp_relabel/{graph}/plain"
        line = {"graph": graph, "N": n, "nnz": e, "F": feat_dim,
                "step_ms_no_reorder": time_ms(lambda: voltrix.spmm(*plain, num_nodes=n, num_edges=e, feat=feat))}
        del plain
        for method in methods:
            info = {}
            for attempt in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h = voltrix.csr_preprocess_reordered(indptr, indices, n, method=method, relabel=True, info=info)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
            fin = voltrix.permute_features(h, feat)
            line[method] = {"picked": h.method, "reorder_plus_preprocess_wall_ms": wall,
                            "step_ms": time_ms(lambda: voltrix.spmm_reordered(h, fin, hash_tag=f"exp_relabel/{graph}/{method}")),
                            "step_ms_unpermuted": time_ms(lambda: voltrix.spmm_reordered(h, fin, unpermute=True)),
                            "two_level": voltrix.two_level_of(h.hspa_packed) is not None,
                            "report": info.get("report")}
            del h, fin
        print(json.dumps(line), flush=True)
        del indptr, indices, feat
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
