"""Time the fused preprocess (sort vs bitmap rank path) on a synthetic config.  usage: prep_time.py [workload] [scale]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "voltrix-spmm_amd"))
import torch
import synth_graphs
import voltrix

wl = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
indptr, indices, _ = synth_graphs.generate(wl, device="cuda", scale=scale)
n = indptr.numel() - 1
print(wl, "N", n, "nnz", indices.numel())
ref = None
for path in ("sort", "bitmap", "auto"):
    if path == "auto":
        os.environ.pop("VOLTRIX_CSR_PATH", None)
    else:
        os.environ["VOLTRIX_CSR_PATH"] = path
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"  {path:7s} ms: " + " ".join(f"{t:.2f}" for t in ts), "T =", int(out[0][-1]))
    if ref is None:
        ref = out
    else:
        for a, b in zip(ref[:3], out[:3]):
            assert torch.equal(a, b), "paths differ"
print("paths agree bit for bit")
