"""A/B of panel kernel builds (harness/experiments/build/panel_*.so), same plan, same process."""
import ctypes, glob, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
import torch, synth_graphs
from voltrix import hybrid
indptr, indices, _ = synth_graphs.generate("reddit_like", device="cuda")
n = indptr.numel() - 1
feat = torch.randn(n, 128, device="cuda").half()
out = torch.zeros(n, 128, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for tau in (4, 3):
    _, _, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, tau)
    print(f"tau {tau}: k-steps {plan.num_ksteps}", flush=True)
    for rep in range(2):
        for path in sorted(glob.glob(os.path.join(HERE, "build", "panel_v*_d*.so"))):
            lib = ctypes.CDLL(path)
            def go():
                assert lib.panel_diag_launch(ctypes.c_void_p(plan.panel_ptr.data_ptr()), ctypes.c_void_p(plan.panel_cols.data_ptr()),
                                             ctypes.c_void_p(plan.panel_bits.data_ptr()), n, 128, ctypes.c_void_p(feat.data_ptr()),
                                             ctypes.c_void_p(out.data_ptr()), 1, ctypes.c_void_p(stream)) == 0
            for _ in range(3): go()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): go()
            b.record(); torch.cuda.synchronize()
            print(f"  {os.path.basename(path):20s} {a.elapsed_time(b) / 10:.3f} ms", flush=True)
