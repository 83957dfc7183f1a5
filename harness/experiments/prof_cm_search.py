"""Phase times of the Cuthill-McKee search on the device (voltrix/reorder.py::bfs_permutation) at BASELINE scale.

    python harness/experiments/prof_cm_search.py [reddit_shuffled|products_shuffled|...] [scale]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]

import torch  # noqa: E402

import synth_graphs  # noqa: E402
from voltrix import capi, reorder  # noqa: E402


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) * 1e3:9.2f} ms", flush=True)
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit_shuffled"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    ip, ix, _ = synth_graphs.generate(name, device="cuda", scale=scale)
    n = ip.numel() - 1
    print(f"{name}: n={n} nnz={ix.numel()}")
    for rep in range(2):
        print(f"--- pass {rep}")
        deg, tie = timed("degrees + tie order", lambda: reorder._cm_degrees(ip, ix, n, n))
        t_ip, t_ix = timed("transpose (radix sort)", lambda: capi.csr_transpose(ip, ix, n, n))
        search = capi.CmSearch(ip, ix, t_ip, t_ix, n, n, tie.to(torch.int32))
        cand = torch.where((search.level < 0) & (deg > 0), tie, torch.full_like(tie, n))
        start = int(torch.argmin(cand))
        nodes, levels = timed("search 1 (probe)", lambda: search.levels(start))
        print(f"    nodes {nodes} levels {levels} level reads {search.syncs}")
        last = search.queue[int(search.level_off[levels - 1]):nodes].long()
        start = int(last[torch.argmin(tie[last])])
        timed("reset levels", lambda: search.level.index_fill_(0, search.queue[:nodes].long(), -1))
        nodes, levels = timed("search 2", lambda: search.levels(start))
        offs = timed("ranks (keys, sorts)", lambda: search.rank_component(levels, 0))
        print(f"    nodes {nodes} levels {levels} sizes {(offs[1:] - offs[:-1]).tolist()[:12]}")
        info = {}
        timed("bfs_permutation, whole", lambda: reorder.bfs_permutation(ip, ix, n, info=info))
        print("   ", info)


if __name__ == "__main__":
    main()
