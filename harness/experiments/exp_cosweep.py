"""How many windows sweep their sorted columns side by side per CU (round 2): the same unit table through tiles of 4, 8 and
more waves per CU (ring depth traded for waves), window format and the two-level residual, alone on the chip.

    python harness/experiments/exp_cosweep.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "voltrix-spmm_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(ROOT, "voltrix-spmm_amd", ".jit_cache"))

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import capi, hybrid  # noqa: E402
from voltrix.schedule import unit_table  # noqa: E402

TILES = [(128, 3, 4), (128, 4, 4), (128, 2, 4), (128, 2, 8), (128, 3, 2), (128, 3, 1), (128, 2, 2), (128, 2, 1)]
if os.environ.get("EXP_TILES") == "fs64":   # 64-column slabs (slab-major): eight waves with a three- or four-deep ring fit
    TILES = [(128, 3, 4), (64, 3, 4), (64, 3, 8), (64, 4, 8), (64, 2, 8), (64, 4, 4)]


def lds(fs, d, w):
    return w * (d * 32 * fs * 2 + (2 * d + 1) * 256)


os.environ.setdefault("VOLTRIX_SLAB_ORDER", "major")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / iters


dev = torch.device("cuda")
indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev)
n, e = indptr.numel() - 1, indices.numel()
F = 128
feat = torch.randn(n, F, device=dev).half()
out = torch.zeros(n, F, device=dev)
stream = torch.cuda.current_stream().cuda_stream
full = voltrix.csr_fused_preprocess_kernel(indptr, indices, n)[:3]
r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
for label, h, ne in (("window format", full, e), ("two-level residual", resid, r_indices.numel())):
    for frac in (1.5, 1.0):
        from voltrix.schedule import default_max_stages

        med = default_max_stages(h[0], n) / 1.5
        tb = unit_table(h[0], n, max(8, int(frac * med)))
        buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
        print(f"{label}: units {tb.num_units} (cut at {frac} x median = {tb.max_stages} stages), cuts {tb.num_cuts}", flush=True)
        for tile in TILES:
            per_cu = min(160 * 1024 // lds(*tile), 8) * tile[2]

            def run():
                assert capi.launch_spmm_sched(h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), n, ne, F, feat.data_ptr(),
                                              out.data_ptr(), tile, stream, 0, 0, False, False, tb, buf.data_ptr()) == 0
                assert capi.launch_combine_partials(tb, buf.data_ptr(), out.data_ptr(), n, F, False, stream) == 0

            print(f"  tile {tile}: LDS {lds(*tile) // 1024:4d} KB per workgroup, <= {per_cu:2d} windows per CU: {timeit(run):.3f} ms", flush=True)
