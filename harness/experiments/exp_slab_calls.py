"""Wide features as ONE operator call (grid over the 128-column slabs) against one call per slab on contiguous slab copies:
is the slab handling inside the launch worth a leading-dimension argument?

    python harness/experiments/exp_slab_calls.py [workload] [F]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402


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
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    ip, ix, _ = synth_graphs.generate(name, device="cuda")
    n, nnz = ip.numel() - 1, ix.numel()
    from voltrix.spmm.spmm import csr_preprocess_device, spmm

    handle = csr_preprocess_device(ip, ix, n)
    handle[1].hash_tag = f"{name}_slab_calls"
    feat = torch.randn(n, feat_dim, device="cuda").half()
    slabs = [feat[:, j:j + 128].contiguous() for j in range(0, feat_dim, 128)]
    whole = lambda: spmm(*handle, num_nodes=n, num_edges=nnz, feat=feat)                       # noqa: E731
    per_slab = lambda: [spmm(*handle, num_nodes=n, num_edges=nnz, feat=s) for s in slabs]     # noqa: E731
    ref = whole()
    got = torch.cat(per_slab(), dim=1)
    print("max abs difference", float((ref - got).abs().max()))
    for _ in range(2):
        print(f"F={feat_dim}: one call {time_ms(whole):.3f} ms   {len(slabs)} calls of 128 columns {time_ms(per_slab):.3f} ms")


if __name__ == "__main__":
    main()
