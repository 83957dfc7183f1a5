"""Panel kernel over pieces of bounded length (round 4; hybrid.panel_parts): the two-level step against the bound on a
piece's k-steps, as a fraction of a CU's fair share S / 256 -- 0 = no table (one workgroup per panel), then 1.0 ... 0.2.
Per graph: k-steps, the longest panel, and per factor the pieces, the partial-tile slots, the step and the two kernels'
own times (events around each launch: they disturb the co-run by a few per cent; the step is timed without them).
    python harness/experiments/exp_panel_parts.py [graph ...]"""
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
from voltrix import hybrid  # noqa: E402
from voltrix.utils import KernelTimer  # noqa: E402


def time_ms(fn, reps=7, batch=5):
    for _ in range(3):
        fn()
    times = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(batch):
            fn()
        e.record()
        e.synchronize()
        times.append(s.elapsed_time(e) / batch)
    return sorted(times)[len(times) // 2]


def main():
    graphs = sys.argv[1:] or ["reddit_sbm", "reddit_like"]
    dev = torch.device("cuda", 0)
    for name in graphs:
        indptr, indices, cfg = synth_graphs.generate(name, device=dev)
        n, e = indptr.numel() - 1, indices.numel()
        feat = torch.randn(n, cfg["feat"], device=dev).half()
        ints = torch.randint(-3, 4, (n, cfg["feat"]), device=dev).half()
        handle = voltrix.csr_preprocess_device(indptr, indices, n)
        handle[1].hash_tag = f"panel_parts/{name}"
        two = voltrix.two_level_of(handle[1])
        plan = two.plan
        nks = torch.diff(plan.panel_ptr)
        print(json.dumps({"graph": name, "N": n, "nnz": e, "F": cfg["feat"], "ksteps": plan.num_ksteps, "panels": plan.num_panels,
                          "longest_panel": int(nks.max()), "median_panel": int(nks.median()),
                          "fair_share_per_cu": plan.num_ksteps / hybrid.NUM_CUS}), flush=True)
        run = lambda x=feat: voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=x)  # noqa: E731
        plan.parts = None
        ref = run(ints).clone()
        for factor in (0.0, 1.0, 0.75, 0.5, 0.35, 0.25, 0.2):
            if factor == 0.0:
                plan.parts = None
            else:
                plan.parts = hybrid.panel_parts(plan.panel_ptr, max(8, int(factor * plan.num_ksteps / hybrid.NUM_CUS)), plan.xcd_ptr)
            same = bool(torch.equal(run(ints), ref))
            ms = time_ms(run)
            with KernelTimer() as timer:
                for _ in range(5):
                    run()
            kernels = {k: round(v[1], 4) for k, v in timer.summary().items()}   # ms per launch
            p = plan.parts
            print(json.dumps({"graph": name, "factor": factor, "cap": p.cap if p else None, "pieces": p.num_parts if p else plan.num_panels,
                              "cut_panels": p.num_cuts if p else 0, "slots": p.num_slots if p else 0,
                              "partial_MB": (p.num_slots * plan.panel_rows * cfg["feat"] * 4 / 1e6) if p else 0.0,
                              "step_ms": round(ms, 4), "integers_bit_equal": same, "kernels_ms": kernels}), flush=True)
        del handle, two, plan


if __name__ == "__main__":
    main()
