"""Operator-level timing: voltrix.spmm on a window-format handle vs a two-level handle (tuner on)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
import torch, synth_graphs, voltrix

name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
f = int(sys.argv[2]) if len(sys.argv) > 2 else 128
indptr, indices, _ = synth_graphs.generate(name, device="cuda")
n, nnz = indptr.numel() - 1, indices.numel()
ip, ix = indptr.cpu(), indices.cpu()
feat = torch.randn(n, f, device="cuda").half()


def timed(handle, tag):
    handle[1].hash_tag = tag
    for _ in range(3):
        out = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        out = voltrix.spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], out


t_w, out_w = timed(voltrix.csr_preprocess(ip, ix, n), f"{name}_window")
t_h, out_h = timed(voltrix.csr_preprocess_hybrid(ip, ix, n), f"{name}_two_level")
rel = float((out_w - out_h).norm() / out_w.norm())
print(f"{name} F={f}: voltrix.spmm window format {t_w:.3f} ms, two-level {t_h:.3f} ms (x{t_w / t_h:.2f}), rel diff {rel:.1e}")
