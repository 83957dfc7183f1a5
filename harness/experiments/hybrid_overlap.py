"""Do the window kernel (residual, gather-bound) and the panel kernel (shared columns, MFMA-bound) overlap when they run
on two streams?  Outputs go to two buffers (no combine pass here).  Usage: hybrid_overlap.py [config] [F]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
import synth_graphs  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.jit_kernels.csr_fused import csr_fused_preprocess_kernel  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit_like"
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    indptr, indices, _ = synth_graphs.generate(name, device="cuda")
    n = indptr.numel() - 1
    feat = torch.randn(n, f, device="cuda").half()
    out = torch.empty(n, f, dtype=torch.float32, device="cuda")
    out2 = torch.empty(n, f, dtype=torch.float32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for waves, rb, tau in [(8, 4, 4), (8, 4, 3), (4, 4, 3)]:
        ri, rx, plan = hybrid.build_panel_plan(indptr, indices, n, None, waves, rb, tau)
        p1, packed, hind, _ = csr_fused_preprocess_kernel(ri, rx, n, n)
        order = torch.empty((n + 15) // 16, dtype=torch.int32, device="cuda")
        capi.launch_window_order(p1, n, order, torch.cuda.current_stream().cuda_stream, 512)
        torch.cuda.synchronize()
        for wtile in [(128, 3, 4), (64, 3, 4), (64, 4, 4)]:
            for depth in (3, 4, 6):
                ptile = (128, depth, 1)

                def win(stream):
                    rc = capi.launch_spmm(p1.data_ptr(), packed.data_ptr(), hind.data_ptr(), n, rx.numel(), f,
                                          feat.data_ptr(), out.data_ptr(), True, wtile, stream.cuda_stream, order.data_ptr())
                    assert rc == 0

                def pan(stream):
                    hybrid.launch_panel(plan, feat, out2, False, tile=ptile, stream=stream.cuda_stream)

                def run(mode):
                    ts = []
                    for it in range(8):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize()
                        a.record()
                        s1.wait_event(a)
                        s2.wait_event(a)
                        if mode in ("both", "win"):
                            win(s1)
                        if mode in ("both", "pan"):
                            pan(s2 if mode == "both" else s1)
                        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
                        e1.record(s1)
                        e2.record(s2)
                        torch.cuda.current_stream().wait_event(e1)
                        torch.cuda.current_stream().wait_event(e2)
                        b.record()
                        torch.cuda.synchronize()
                        ts.append(a.elapsed_time(b))
                    ts = sorted(ts[2:])
                    return ts[len(ts) // 2]

                try:
                    tw, tp, tb = run("win"), run("pan"), run("both")
                except Exception as e:
                    print(f"  panel{plan.panel_rows} tau{tau} wtile{wtile} pdepth{depth}: {e}")
                    continue
                print(f"  panel{plan.panel_rows} tau{tau} wtile{wtile} pdepth{depth}: window {tw:.3f} panel {tp:.3f} "
                      f"sum {tw + tp:.3f} concurrent {tb:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
