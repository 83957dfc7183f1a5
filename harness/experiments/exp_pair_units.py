"""Experiment (round 2): two units per wave in the window kernel (spmm_tc16_kernel<T, 2>, harness/experiments/pair_units.hip)
against one unit per wave, same unit table: window format and two-level residual of the reddit-like graph, alone and beside
the panel kernel.  Build the .so first (hipcc, see the command in the log header); run on the GPU box."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.schedule import unit_table  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, "build", "pair_units.so"))
dev = torch.device("cuda")
indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev)
n, e, F = indptr.numel() - 1, indices.numel(), 128
feat = torch.randn(n, F, device=dev).half()
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
full = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
out = torch.zeros(n, F, device=dev)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / iters


from voltrix.schedule import default_max_stages  # noqa: E402

CASES = [("window format", full, 1.5), ("two-level residual", resid, 1.5), ("two-level residual", resid, 1.0),
         ("two-level residual", resid, 1.25), ("two-level residual", resid, 2.0), ("two-level residual", resid, 3.0)]
for label, h, factor in CASES:
    tb = unit_table(h[0], n, max(8, int(factor * default_max_stages(h[0], n) / 1.5)))
    label = f"{label} (units cut at {factor} x median = {tb.max_stages} stages)"
    buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)

    def launch(stream, atomic, nu):
        rc = lib.pair_units_launch(ctypes.c_void_p(h[0].data_ptr()), ctypes.c_void_p(h[1].data_ptr()),
                                   ctypes.c_void_p(h[2].data_ptr()), n, F, ctypes.c_void_p(feat.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), int(atomic), ctypes.c_void_p(tb.units.data_ptr()),
                                   ctypes.c_void_p(tb.unit_ptr.data_ptr()), tb.max_units_per_xcd,
                                   ctypes.c_void_p(buf.data_ptr()), nu, ctypes.c_void_p(stream))
        assert rc == 0, rc

    def whole(nu):
        launch(main.cuda_stream, False, nu)
        assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, F, False, main.cuda_stream) == 0

    res = {}
    for nu in (1, 2):
        out.fill_(float("nan"))
        whole(nu)
        torch.cuda.synchronize()
        res[nu] = out.clone()
    print(f"{label}: units {tb.num_units} | results of 1 and 2 units per wave bit-equal {torch.equal(res[1], res[2])}, NaNs "
          f"{int(torch.isnan(res[2]).sum())} | alone: 1 unit/wave {timed(lambda: whole(1)):.3f} ms, 2 units/wave "
          f"{timed(lambda: whole(2)):.3f} ms", flush=True)
    if label.startswith("two"):  # noqa: E501
        def pair(nu):
            def go():
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    assert capi.launch_spmm_panel(plan, feat.data_ptr(), out.data_ptr(), F, 2, False, (128, 3, 1), 0,
                                                  side.cuda_stream) == 0
                launch(main.cuda_stream, True, nu)
                main.wait_stream(side)
            return go
        print(f"  beside the panel kernel: 1 unit/wave {timed(pair(1)):.3f} ms, 2 units/wave {timed(pair(2)):.3f} ms", flush=True)


# ---- threshold of the two-level split with two units per wave (the balance between the two kernels moves) -----------------
print("tau sweep, pair = panel kernel || residual window kernel (atomic), units cut at 1.25 x median", flush=True)
for tau in (2, 3, 4):
    r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, tau)
    h = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
    tb = unit_table(h[0], n, max(8, int(1.25 * default_max_stages(h[0], n) / 1.5)))
    buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)

    def launch(stream, atomic, nu):
        rc = lib.pair_units_launch(ctypes.c_void_p(h[0].data_ptr()), ctypes.c_void_p(h[1].data_ptr()),
                                   ctypes.c_void_p(h[2].data_ptr()), n, F, ctypes.c_void_p(feat.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), int(atomic), ctypes.c_void_p(tb.units.data_ptr()),
                                   ctypes.c_void_p(tb.unit_ptr.data_ptr()), tb.max_units_per_xcd,
                                   ctypes.c_void_p(buf.data_ptr()), nu, ctypes.c_void_p(stream))
        assert rc == 0, rc

    def pair(nu):
        def go():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                assert capi.launch_spmm_panel(plan, feat.data_ptr(), out.data_ptr(), F, 2, False, (128, 3, 1), 0,
                                              side.cuda_stream) == 0
            launch(main.cuda_stream, True, nu)
            main.wait_stream(side)
        return go
    print(f"  tau {tau}: shared edges {plan.num_shared_edges} k-steps {plan.num_ksteps} | pair with 1 unit/wave "
          f"{timed(pair(1)):.3f} ms, 2 units/wave {timed(pair(2)):.3f} ms", flush=True)
