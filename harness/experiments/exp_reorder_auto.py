"""method="auto" at BASELINE scale: what the statistics say about every candidate, what is picked, what
the reorder costs (wall clock, second call) and the operator's step with and without it.
    python harness/experiments/exp_reorder_auto.py reddit_shuffled,reddit_like,products_shuffled,reddit_sbm_shuffled [feat]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")

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
    graphs = (sys.argv[1] if len(sys.argv) > 1 else "reddit_shuffled,reddit_like,products_shuffled").split(",")
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda", 0)
    for graph in graphs:
        indptr, indices, _ = synth_graphs.generate(graph, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, feat_dim, device=dev).half()
        plain = voltrix.csr_preprocess_device(indptr, indices, n)
        plain[1].hash_tag = f"exp_reorder_auto/{graph}/plain"
        t_plain = time_ms(lambda: voltrix.spmm(*plain, num_nodes=n, num_edges=e, feat=feat))
        for attempt in range(2):       # the second call: kernels loaded, allocator warm
            info = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            handle = voltrix.csr_preprocess_reordered(indptr, indices, n, info=info)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        t_auto = time_ms(lambda: voltrix.spmm_reordered(handle, feat, hash_tag=f"exp_reorder_auto/{graph}/auto"))
        print(json.dumps({"graph": graph, "N": n, "nnz": e, "F": feat_dim, "picked": info["picked"],
                          "reorder_plus_preprocess_wall_ms": wall, "step_ms_no_reorder": t_plain, "step_ms_auto": t_auto,
                          "report": info["report"]}), flush=True)
        del handle, plain, indptr, indices, feat
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
