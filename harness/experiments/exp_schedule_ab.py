"""A/B of the two round-4 schedules of the two-level step -- XCD ranges of equal work (hybrid.balance_xcd_ranges) and panels
in pieces (hybrid.panel_parts) -- on natural, block-model and spectrally reordered graphs: each on / off, same handle, same
bits (checked on integers).  `name+spectral` = the label-shuffled stand-in put back in order by reorder.spectral_permutation.
    python harness/experiments/exp_schedule_ab.py [graph ...]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("VOLTRIX_TUNE_SPACE", "none")
os.environ["VOLTRIX_HYBRID"] = "1"
os.environ["VOLTRIX_HYBRID_MIN_SHARE"] = "0"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid, reorder  # noqa: E402
from voltrix.utils import KernelTimer  # noqa: E402

from exp_panel_parts import time_ms  # noqa: E402


def main():
    graphs = sys.argv[1:] or ["reddit_shuffled+spectral", "reddit_sbm_shuffled+spectral", "reddit_sbm", "reddit_like"]
    dev = torch.device("cuda", 0)
    for name in graphs:
        base, _, order = name.partition("+")
        indptr, indices, cfg = synth_graphs.generate(base, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        if order == "spectral":
            perm = reorder.spectral_permutation(indptr, indices, n, iterations=12)
            indptr, indices = reorder.permute_rows_csr(indptr, indices, n, perm)
        feat = torch.randn(n, cfg["feat"], device=dev).half()
        ints = torch.randint(-3, 4, (n, cfg["feat"]), device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"schedule_ab/{name}"
        two = voltrix.two_level_of(handle[1])
        plan = two.plan
        balanced = (plan.xcd_ptr, plan.max_panels_per_xcd, plan.panel_order, two.window_xcd_ptr)
        equal = (None, 0, hybrid.longest_first_order(plan.panel_ptr), None)
        nks = torch.diff(plan.panel_ptr)
        print(json.dumps({"graph": name, "ksteps": plan.num_ksteps, "longest_panel": int(nks.max()), "median_panel": int(nks.median()),
                          "fair_share_per_cu": plan.num_ksteps / hybrid.NUM_CUS,
                          "panels_over_0.75_share": int((nks > 0.75 * plan.num_ksteps / hybrid.NUM_CUS).sum())}), flush=True)
        run = lambda x=feat: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)  # noqa: E731
        ref = None
        for label, ranges, factor in (("equal counts, whole panels", equal, 0.0), ("equal work, whole panels", balanced, 0.0),
                                      ("equal work, pieces 1.0", balanced, 1.0), ("equal work, pieces 0.75", balanced, 0.75),
                                      ("equal counts, pieces 0.75", equal, 0.75), ("equal work, pieces 1.25", balanced, 1.25),
                                      ("equal work, pieces 1.5", balanced, 1.5)):
            plan.xcd_ptr, plan.max_panels_per_xcd, plan.panel_order, two.window_xcd_ptr = ranges
            plan.parts = None
            if factor > 0:
                plan.parts = hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)), plan.xcd_ptr)
            out = run(ints).clone()
            ref = out if ref is None else ref
            same = bool(torch.equal(out, ref))
            ms = time_ms(run)
            with KernelTimer() as timer:
                for _ in range(5):
                    run()
            kernels = {k: round(v[1], 4) for k, v in timer.summary().items()}
            p = plan.parts
            print(json.dumps({"graph": name, "variant": label, "cut_panels": p.num_cuts if p else 0, "slots": p.num_slots if p else 0,
                              "step_ms": round(ms, 4), "integers_bit_equal": same, "kernels_ms": kernels}), flush=True)
        del handle, two, plan


if __name__ == "__main__":
    main()
