"""Two-level format on the headline graph: window kernel on the residual + panel kernel on the shared columns.
Usage: python harness/experiments/hybrid_time.py [config] [F]"""
import itertools
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
import synth_graphs  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.jit_kernels.csr_fused import csr_fused_preprocess_kernel  # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    indptr, indices, _ = synth_graphs.generate(name, device="cuda")
    n, nnz = indptr.numel() - 1, indices.numel()
    feat = torch.randn(n, f, device="cuda").half()
    out = torch.empty(n, f, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    wtile = (min(128, max(32, f)), 3, 4)

    def window_time(ip, ix):
        p1, packed, hind, _ = csr_fused_preprocess_kernel(ip, ix, n, n)
        order = torch.empty((n + 15) // 16, dtype=torch.int32, device="cuda")
        best = None
        for chunk in (128, 512, 2048):
            capi.launch_window_order(p1, n, order, stream, chunk)
            t = timed(lambda: capi.launch_spmm(p1.data_ptr(), packed.data_ptr(), hind.data_ptr(), n, ix.numel(), f,
                                               feat.data_ptr(), out.data_ptr(), True, wtile, stream, order.data_ptr()))
            best = t if best is None or t < best else best
        return best, int(p1[-1])

    t_full, blocks = window_time(indptr, indices)
    print(f"{name} N={n} nnz={nnz} F={f}: window kernel alone {t_full:.3f} ms ({blocks} TC blocks)", flush=True)
    for waves, rb, tau in [(8, 4, 3), (8, 4, 2), (8, 4, 4), (4, 4, 3), (4, 4, 2), (8, 2, 3)]:
        t0 = time.time()
        ri, rx, plan = hybrid.build_panel_plan(indptr, indices, n, None, waves, rb, tau)
        torch.cuda.synchronize()
        t_plan = time.time() - t0
        t_res, rblocks = window_time(ri, rx)
        line = (f"  panel {plan.panel_rows} rows (waves {waves}, rb {rb}) tau {tau}: shared {plan.num_shared_edges / nnz:.1%} "
                f"of edges in {plan.num_ksteps} k-steps; residual {t_res:.3f} ms "
                f"({rblocks} blocks); plan {t_plan * 1e3:.0f} ms; panel:")
        for depth in (4, 6, 8):
            tile = (128 if f >= 128 else f, depth, 1 if f >= 128 else 2)  # (fs, depth, ksteps)
            try:
                t_p = timed(lambda: hybrid.launch_panel(plan, feat, out, True, tile=tile))
            except Exception as e:  # tile not instantiated
                line += f" d{depth} n/a"
                continue
            line += f" d{depth} {t_p:.3f}"
        print(line + f" ms -> total {t_res:.3f} + best", flush=True)


if __name__ == "__main__":
    main()
