"""The residual's window kernel WITHOUT a panel kernel beside it, in the two output modes the two-level step can use: plain
stores (every row written) and float atomics onto a zero-filled C (the pair's join), plus the zero fill and the combine pass;
and the panel kernel alone in its three output modes.  What does the join itself cost?
    python harness/experiments/exp_window_alone.py [graph ...]"""
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
        n, e = indptr.numel() - 1, indices.numel()
        f = cfg["feat"]
        feat = torch.randn(n, f, device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"window_alone/{name}"
        two = voltrix.two_level_of(handle[1])
        out = torch.zeros(n, f, device=dev)
        voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)     # tags the residual, builds tables
        line = {"graph": name, "pair_ms": round(time_ms(lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)), 4)}

        def window(atomic, beside):
            pending = spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=two.plan.num_resid_edges,
                                  embedding_dim=f, input=feat, output=out, atomic_out=atomic, beside_panel=beside,
                                  defer_combine=True, xcd_ptr=two.window_xcd_ptr)
            if pending is not None:
                pending.run()

        for atomic in (False, True):
            for beside in (False, True):
                line[f"window_alone_{'atomic' if atomic else 'store'}_{'beside_tile' if beside else 'own_tile'}_ms"] = \
                    round(time_ms(lambda: window(atomic, beside)), 4)
        line["zero_fill_ms"] = round(time_ms(lambda: out.zero_()), 4)
        for acc in (0, 1, 2):
            line[f"panel_alone_accumulate{acc}_ms"] = round(time_ms(lambda: hybrid.launch_panel(two.plan, feat, out, accumulate=acc)), 4)
        print(json.dumps(line), flush=True)
        del handle, two


if __name__ == "__main__":
    main()
