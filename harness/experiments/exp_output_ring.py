"""Does it matter that back-to-back operator calls write the SAME output buffer (the caching allocator hands the freed block
back) -- zero fill and float atomics onto lines the previous call left dirty in L2 / the Infinity Cache?  And the same B?

    python harness/experiments/exp_output_ring.py [workload] [F]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
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


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    feat_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ip, ix, _ = synth_graphs.generate(name, device="cuda")
    n, nnz = ip.numel() - 1, ix.numel()
    handle = csr_preprocess_device(ip, ix, n)
    handle[1].hash_tag = f"{name}_output_ring"
    feats = [torch.randn(n, feat_dim, device="cuda").half() for _ in range(4)]
    ring = []

    def same():
        spmm(*handle, num_nodes=n, num_edges=nnz, feat=feats[0])

    def ring_out(depth):
        def fn():
            ring.append(spmm(*handle, num_nodes=n, num_edges=nnz, feat=feats[0]))
            if len(ring) > depth:
                ring.pop(0)
        return fn

    state = {"i": 0}

    def ring_b():
        state["i"] = (state["i"] + 1) % 4
        spmm(*handle, num_nodes=n, num_edges=nnz, feat=feats[state["i"]])

    for _ in range(2):
        print(f"same B, output block reused      {time_ms(same):.4f} ms")
        for depth in (1, 3):
            ring.clear()
            print(f"same B, {depth + 1} output blocks in turn   {time_ms(ring_out(depth)):.4f} ms")
        ring.clear()
        print(f"4 B operands in turn, output reused {time_ms(ring_b):.4f} ms")


if __name__ == "__main__":
    main()
