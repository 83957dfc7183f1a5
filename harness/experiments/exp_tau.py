"""tau (how many rows of a 512-row panel must share a column for it to go to the panel plan) under the round-4 schedules:
shared fraction, k-steps, both kernels' times and the step, per graph.  Also the estimate the reorder rule would make
(reorder.order_statistics' model) next to the measurement.
    python harness/experiments/exp_tau.py [graph ...]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid, reorder  # noqa: E402
from voltrix.utils import KernelTimer  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    graphs = sys.argv[1:] or ["reddit_like", "reddit_sbm"]
    dev = torch.device("cuda", 0)
    for name in graphs:
        indptr, indices, cfg = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, cfg["feat"], device=dev).half()
        indptr_c, indices_c = indptr.cpu(), indices.cpu()
        for tau in (2, 3, 4, 5, 6, 8):
            two = voltrix.csr_preprocess_hybrid(indptr_c, indices_c, n, tau=tau)
            two.hash_tag = f"tau/{name}/{tau}"
            plan = two.plan
            run = lambda: voltrix.spmm_two_level(two, feat)  # noqa: E731
            ms = time_ms(run)
            with KernelTimer() as timer:
                for _ in range(5):
                    run()
            resid_edges = plan.num_resid_edges
            est = reorder.AUTO_MS_JOIN + max(reorder.AUTO_MS_PER_RESIDUAL_BLOCK * resid_edges / 8.0,
                                             reorder.AUTO_MS_PER_KSTEP_PER_CU * plan.num_ksteps / hybrid.NUM_CUS)
            print(json.dumps({"graph": name, "tau": tau, "shared_fraction": round(plan.num_shared_edges / e, 4), "ksteps": plan.num_ksteps,
                              "residual_tc_blocks": int(two.blk_offsets[-1]), "longest_panel": int(torch.diff(plan.panel_ptr).max()),
                              "cut_panels": plan.parts.num_cuts if plan.parts else 0, "step_ms": round(ms, 4), "estimate_ms": round(est, 3),
                              "kernels_ms": {k: round(v[1], 4) for k, v in timer.summary().items()}}), flush=True)
            del two, plan


if __name__ == "__main__":
    main()
