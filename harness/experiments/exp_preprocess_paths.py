"""Fused preprocess: wall time of count + fill per rank algorithm (VOLTRIX_CSR_PATH) on the BASELINE stand-ins (round 2).

    python harness/experiments/exp_preprocess_paths.py [workload ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "voltrix-spmm_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(ROOT, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402

for name in sys.argv[1:] or ["reddit_like", "products_like", "powerlaw_4m"]:
    indptr, indices, _ = synth_graphs.generate(name, device=torch.device("cuda"))
    n, e = indptr.numel() - 1, indices.numel()
    per_window = (indptr[16::16] - indptr[:-16:16]).float()
    print(f"{name}: N={n} nnz={e} edges per window: median {int(per_window.median())} max {int(per_window.max())}", flush=True)
    ref = None
    paths = os.environ.get("EXP_PATHS", "auto,sort,bitmap,mixed").split(",")
    for path in paths:
        if path == "auto":
            os.environ.pop("VOLTRIX_CSR_PATH", None)
        else:
            os.environ["VOLTRIX_CSR_PATH"] = path
        if name in ("powerlaw_4m", "papers_like") and path in ("sort", "bitmap") and path != "auto" and len(paths) > 1:
            print("  sort  : skipped (global-memory bitonic sort of 100 k-edge windows)", flush=True)
            continue
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        sums = tuple(int(t.view(torch.int32).long().sum()) for t in h)
        ref = ref or sums
        print(f"  {path:6s}: {min(times):8.2f} ms (count + fill + host sync), handle checksums equal {sums == ref}", flush=True)
        del h
    del indptr, indices
    torch.cuda.empty_cache()
