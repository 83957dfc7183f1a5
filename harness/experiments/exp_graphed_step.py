"""The headline step eager (the drop-in operator call) against a replayed HIP graph of it (voltrix.GraphedSpMM).
    python harness/experiments/exp_graphed_step.py [workload] [F]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix.spmm.spmm import csr_preprocess_device, spmm  # noqa: E402


def time_ms(fn, iters=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ip, ix, _ = synth_graphs.generate(name, device="cuda")
n, nnz = ip.numel() - 1, ix.numel()
handle = csr_preprocess_device(ip, ix, n)
handle[1].hash_tag = f"{name}_graphed_step"
feat = torch.randn(n, feat_dim, device="cuda").half()
eager = lambda: spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)   # noqa: E731
eager()
op = voltrix.GraphedSpMM(*handle, n, nnz, feat)
for _ in range(2):
    print(f"{name} F={feat_dim}: eager {time_ms(eager):.4f} ms   graph replay {time_ms(lambda: op(feat)):.4f} ms")
