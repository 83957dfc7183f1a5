"""Panel kernel ring depth beside the window kernel, now that the shared metadata slots freed LDS: (128, 3) = 36 KiB, (128, 4) =
48 KiB -- both fit a CU beside the (128, 3, 4) window workgroup (103 KiB).
    python harness/experiments/exp_panel_depth.py"""
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
from voltrix.utils import KernelTimer  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    shipped = hybrid.default_panel_tile
    for name in ("reddit_like", "reddit_sbm"):
        indptr, indices, cfg = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, cfg["feat"], device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"panel_depth/{name}"
        run = lambda: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)  # noqa: E731
        ref = run().clone()
        for _ in range(2):
            for depth in (3, 4):
                hybrid.default_panel_tile = (lambda d: (lambda f, w, rb=4: (128, d, 1)))(depth)
                same = bool(torch.equal(run(), ref))
                ms = time_ms(run)
                with KernelTimer() as timer:
                    for _ in range(5):
                        run()
                print(json.dumps({"graph": name, "panel_depth": depth, "step_ms": round(ms, 4), "bits_equal": same,
                                  "kernels_ms": {k: round(v[1], 4) for k, v in timer.summary().items()}}), flush=True)
        hybrid.default_panel_tile = shipped
        del handle


if __name__ == "__main__":
    main()
