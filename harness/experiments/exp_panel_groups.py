"""Experiment (round 2): launch order of the panel kernel -- longest first per XCD range (shipped) against longest-first GROUPS of
g consecutive panels (neighbours share their band columns; launched side by side they can share the gathered rows through L2).
Pair = panel kernel || residual window kernel (two units per wave), reddit-like F=128.

    python harness/experiments/exp_panel_groups.py
"""
import dataclasses
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
from voltrix.schedule import default_max_stages, unit_table  # noqa: E402

dev = torch.device("cuda")
indptr, indices, _ = synth_graphs.generate("reddit_like", device=dev)
n, F = indptr.numel() - 1, 128
feat = torch.randn(n, F, device=dev).half()
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
r_indptr, r_indices, plan = hybrid.build_panel_plan(indptr, indices, n, None, 8, 4, 3)
rn = r_indices.numel()
resid = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, n)[:3]
tb = unit_table(resid[0], n, max(8, int(1.25 * default_max_stages(resid[0], n) / 1.5)))
buf = torch.empty(max(1, tb.num_slots) * 16 * F, dtype=torch.float32, device=dev)
out = torch.zeros(n, F, device=dev)
nks = (plan.panel_ptr[1:] - plan.panel_ptr[:-1]).long()
num_panels = nks.numel()
per_xcd = (num_panels + 7) // 8


def grouped_order(g):
    """inside every XCD range: groups of g consecutive panels, groups by descending total length, panels of a group in order"""
    idx = torch.arange(num_panels, device=dev)
    xcd = idx // per_xcd
    grp = (idx - xcd * per_xcd) // g
    gid = xcd * (per_xcd // g + 2) + grp
    gsum = torch.zeros(int(gid.max()) + 1, dtype=torch.int64, device=dev).index_add_(0, gid, nks)
    top = int(gsum.max())
    key = (xcd * (top + 1) + (top - gsum[gid])) * (per_xcd + 1) + (idx - xcd * per_xcd)
    return torch.argsort(key, stable=True).to(torch.int32)


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


def window(stream):
    assert capi.launch_spmm_sched(resid[0].data_ptr(), resid[1].data_ptr(), resid[2].data_ptr(), n, rn, F, feat.data_ptr(),
                                  out.data_ptr(), (128, 3, 4), stream, 0, 0, True, False, tb, buf.data_ptr(), 0, 2) == 0


orders = [("longest first (shipped)", plan.panel_order), ("natural", None)] + [(f"groups of {g}", grouped_order(g)) for g in (2, 4, 8, 16)]
for label, order in orders:
    p = dataclasses.replace(plan, panel_order=order)

    def panel(stream, p=p):
        assert capi.launch_spmm_panel(p, feat.data_ptr(), out.data_ptr(), F, 2, False, (128, 3, 1), 0, stream) == 0

    def pair():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            panel(side.cuda_stream)
        window(main.cuda_stream)
        main.wait_stream(side)

    print(f"panel order {label:24s}: panel kernel alone {timed(lambda: panel(main.cuda_stream)):.3f} ms | pair {timed(pair):.3f} ms",
          flush=True)
