"""The join of the two-level step without float atomics: the window kernel STORES C (every row; no zero fill), the panel kernel
STORES its tiles to a second buffer P on the side stream, and after the join one pass adds P onto C.  Same bits (two addends per
element either way).  exp_corun_diag2 suggested that the panel kernel's atomic epilogue costs the window kernel more than it
costs the panel kernel itself; this prices the alternative.
    python harness/experiments/exp_two_buffers.py [graph ...]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_HYBRID"] = "1"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid  # noqa: E402
from voltrix.jit_kernels.spmm import spmm_kernel  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    graphs = sys.argv[1:] or ["reddit_like", "reddit_sbm"]
    dev = torch.device("cuda", 0)
    for name in graphs:
        indptr, indices, cfg = synth_graphs.generate(name, device=dev)
        n, e, f = indptr.numel() - 1, indices.numel(), cfg["feat"]
        feat = torch.randn(n, f, device=dev).half()
        ints = torch.randint(-3, 4, (n, f), device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"two_buffers/{name}"
        two = voltrix.two_level_of(handle[1])
        plan = two.plan
        side = hybrid.side_stream(dev)
        tile_buffer = torch.empty(n, f, device=dev)

        def atomic_join(x):
            return voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)

        def two_buffers(x):
            out = torch.empty(n, f, device=dev)
            main_stream = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main_stream)
            side.wait_event(fork)
            pending_panel = hybrid.launch_panel(plan, x, tile_buffer, accumulate=0, stream=side.cuda_stream, defer_combine=True)
            join = torch.cuda.Event()
            join.record(side)
            pending = spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=plan.num_resid_edges,
                                  embedding_dim=f, input=x, output=out, atomic_out=False, beside_panel=True, defer_combine=True,
                                  xcd_ptr=two.window_xcd_ptr)
            if pending is not None:
                pending.run()                      # cut windows: their rows = the sum of their partial tiles (store)
            main_stream.wait_event(join)
            if pending_panel is not None:
                pending_panel.accumulate = 0
                pending_panel.run()                # cut panels: rows of the tile buffer = the sum of the pieces
            out.add_(tile_buffer)
            return out

        same = bool(torch.equal(two_buffers(ints), atomic_join(ints)))
        rel = float((two_buffers(feat) - atomic_join(feat)).abs().max())
        print(json.dumps({"graph": name, "atomic_join_ms": round(time_ms(lambda: atomic_join(feat)), 4),
                          "two_buffers_ms": round(time_ms(lambda: two_buffers(feat)), 4), "integers_bit_equal": same,
                          "random_max_abs_diff": rel}), flush=True)
        del handle, two


if __name__ == "__main__":
    main()
