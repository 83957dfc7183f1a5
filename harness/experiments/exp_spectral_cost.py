"""Cost against quality of the spectral row order (VERDICT r3 item 4 asks for <= 80 ms on the reddit-size graph; round 4's
default costs 111 ms with method="auto"): subspace steps x refinement passes -> phase times, and what the order does to the
format (TC blocks, shared fraction, k-steps, the auto rule's estimated step) by reorder.order_statistics.
    python harness/experiments/exp_spectral_cost.py [graph ...]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
from voltrix import reorder  # noqa: E402


def main():
    graphs = sys.argv[1:] or ["reddit_shuffled"]
    dev = torch.device("cuda", 0)
    for name in graphs:
        indptr, indices, _ = synth_graphs.generate(name, device=dev)
        n = indptr.numel() - 1
        reorder.spectral_permutation(indptr, indices, n, iterations=2, refine=1)     # kernel load, unit tables of the handles' shapes
        base = reorder.order_statistics(indptr, indices, n)
        print(json.dumps({"graph": name, "order": "identity", **base}), flush=True)
        for vectors, iterations, refine in ((32, 16, 4), (32, 12, 4), (32, 12, 3), (32, 8, 4), (32, 8, 3), (32, 8, 2), (32, 6, 3),
                                            (16, 12, 3), (16, 8, 3)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            perm = reorder.spectral_permutation(indptr, indices, n, vectors=vectors, iterations=iterations, refine=refine)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            _, info = reorder.spectral_permutation(indptr, indices, n, vectors=vectors, iterations=iterations, refine=refine,
                                                   return_info=True)
            p_indptr, p_indices = reorder.permute_rows_csr(indptr, indices, n, perm)
            st = reorder.order_statistics(p_indptr, p_indices, n)
            print(json.dumps({"graph": name, "vectors": vectors, "iterations": iterations, "refine": refine, "wall_ms": round(wall, 1),
                              "phase_ms": info["phase_ms"], **st}), flush=True)
            del perm, p_indptr, p_indices


if __name__ == "__main__":
    main()
