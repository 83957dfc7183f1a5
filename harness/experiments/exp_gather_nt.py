"""Non-temporal row gathers (aux = 2 on the LDS-DMA) on the HBM-resident configurations (round 2 experiment; needs the
experiment build of spmm_kernels.hpp that reads VOLTRIX_GATHER_NT -- `git log -S gather_nt` -- the shipped kernel has no such
flag: the result was 20-70 % SLOWER everywhere, profiles/r02/experiment_gather_nt.log).

    python harness/experiments/exp_gather_nt.py [workload:F ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "voltrix-spmm_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(ROOT, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi  # noqa: E402


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / iters


dev = torch.device("cuda")
stream = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:] or ["reddit_like:128", "products_like:128", "products_like:512", "papers_like:128", "powerlaw_4m:256"]:
    name, F = spec.split(":")
    F = int(F)
    indptr, indices, _ = synth_graphs.generate(name, device=dev)
    n, e = indptr.numel() - 1, indices.numel()
    h = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
    del indptr, indices
    feat = torch.randn(n, F, device=dev).half()
    out = torch.empty(n, F, device=dev)
    for tile in ((128, 3, 4), (128, 3, 1), (64, 3, 1), (64, 4, 1)):
        res = []
        for nt in (False, True):
            if nt:
                os.environ["VOLTRIX_GATHER_NT"] = "1"
            else:
                os.environ.pop("VOLTRIX_GATHER_NT", None)
            res.append(timeit(lambda: capi.launch_spmm_sched(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, e, F,
                                                             feat.data_ptr(), out.data_ptr(), tile, stream)))
        print(f"{name} F={F} tile {tile}: gathers temporal {res[0]:.3f} ms, non-temporal {res[1]:.3f} ms", flush=True)
    del h, feat, out
    torch.cuda.empty_cache()
