"""End-to-end two-level SpMM on the headline graph: window kernel (residual) on the main stream, panel kernel on a side
stream into a second buffer, add pass -- against the window kernel alone.  Usage: hybrid_e2e.py [config] [F]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
import synth_graphs  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.jit_kernels.csr_fused import csr_fused_preprocess_kernel  # noqa: E402


def timed(fn, iters=12, warm=4):
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
    ncols = indptr.numel() - 1
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0   # < 1: the first rows only = one rank's shard of a multi-GPU run
    if frac < 1.0:
        keep = int(ncols * frac) // 16 * 16
        indptr = indptr[: keep + 1].clone()
        indices = indices[: int(indptr[-1])].clone()
    n, nnz = indptr.numel() - 1, indices.numel()
    feat = torch.randn(ncols, f, device="cuda").half()
    out = torch.empty(n, f, dtype=torch.float32, device="cuda")
    out2 = torch.empty(n, f, dtype=torch.float32, device="cuda")
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    fs = min(128, max(32, f))

    def handle_of(ip, ix):
        p1, packed, hind, _ = csr_fused_preprocess_kernel(ip, ix, n, ncols)
        orders = {}
        for chunk in (128, 512, 2048):
            o = torch.empty((n + 15) // 16, dtype=torch.int32, device="cuda")
            capi.launch_window_order(p1, n, o, main_s.cuda_stream, chunk)
            orders[chunk] = o
        return p1, packed, hind, orders

    def window(h, ix_n, tile, chunk, dst):
        rc = capi.launch_spmm(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, ix_n, f, feat.data_ptr(), dst.data_ptr(),
                              True, tile, main_s.cuda_stream, h[3][chunk].data_ptr())
        assert rc == 0

    full = handle_of(indptr, indices)
    base = min(timed(lambda: window(full, nnz, (fs, 3, 4), c, out)) for c in (128, 512, 2048))
    print(f"{name} N={n} nnz={nnz} F={f}: window kernel alone {base:.3f} ms", flush=True)
    ref = out.clone()
    for waves, rb in ((8, 4),):
        for tau in (3, 4):
            ri, rx, plan = hybrid.build_panel_plan(indptr, indices, n, ncols, waves, rb, tau)
            h = handle_of(ri, rx)
            for wtile in ((fs, 3, 4),):
                for chunk in (512, 2048):
                    for pdepth in (3,):
                        ptile = (fs, pdepth, 1 if fs == 128 else 2)

                        def two_level():
                            fork = torch.cuda.Event()
                            fork.record(main_s)
                            side.wait_event(fork)
                            hybrid.launch_panel(plan, feat, out2, False, tile=ptile, stream=side.cuda_stream)
                            join = torch.cuda.Event()
                            join.record(side)
                            window(h, rx.numel(), wtile, chunk, out)
                            main_s.wait_event(join)
                            capi.launch_add_inplace_f32(out, out2, main_s.cuda_stream)

                        def sequential():
                            window(h, rx.numel(), wtile, chunk, out)
                            hybrid.launch_panel(plan, feat, out, True, tile=ptile)

                        try:
                            t2, t1 = timed(two_level), timed(sequential)
                        except Exception as e:
                            print(f"   tau {tau} {wtile} pdepth {pdepth}: {e}")
                            continue
                        two_level()
                        torch.cuda.synchronize()
                        err = float((out - ref).norm() / ref.norm())
                        print(f"  panel {plan.panel_rows} tau {tau} shared {plan.num_shared_edges / nnz:.1%} ksteps "
                              f"{plan.num_ksteps} | wtile {wtile} chunk {chunk} pdepth {pdepth}: two streams {t2:.3f} ms, "
                              f"one stream {t1:.3f} ms  (x{base / t2:.2f}, rel diff {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
