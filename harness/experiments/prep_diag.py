"""Kernel-level timing of the two fused-preprocess launches (HIP events).
usage: prep_diag.py [workload] [sort|bitmap|auto]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "voltrix-spmm_amd"))
import torch, synth_graphs, voltrix
from voltrix import capi
wl = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
indptr, indices, _ = synth_graphs.generate(wl, device="cuda", scale=1.0)
n = indptr.numel() - 1
W = (n + 15) // 16
if len(sys.argv) > 2 and sys.argv[2] != "auto":
    os.environ["VOLTRIX_CSR_PATH"] = sys.argv[2]
stream = torch.cuda.current_stream().cuda_stream
ws = torch.empty(capi.csr_preprocess_workspace_bytes(n, n, indices.numel()), dtype=torch.uint8, device="cuda")
bp = torch.empty(W, dtype=torch.int32, device="cuda"); p1 = torch.empty(W + 1, dtype=torch.int32, device="cuda")
st = torch.empty(1, dtype=torch.int32, device="cuda")
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
tc = timed(lambda: capi.launch_csr_window_count(indptr, indices, n, n, ws, bp, p1, st, stream))
T = int(p1[-1]); print(wl, "N", n, "nnz", indices.numel(), "T", T, "count ms %.3f" % tc)
packed = torch.empty(T * 4, dtype=torch.uint32, device="cuda"); hind = torch.empty(T * 8, dtype=torch.int32, device="cuda")
print("fill ms %.3f" % timed(lambda: capi.launch_csr_fill(indptr, indices, n, n, ws, p1, packed, hind, stream)))
